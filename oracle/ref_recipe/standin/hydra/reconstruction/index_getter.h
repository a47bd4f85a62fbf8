// stand-in for <hydra/reconstruction/index_getter.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../../ref_standin.h"
