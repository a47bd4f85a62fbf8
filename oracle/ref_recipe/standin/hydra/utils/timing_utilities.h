// stand-in for <hydra/utils/timing_utilities.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../../ref_standin.h"
