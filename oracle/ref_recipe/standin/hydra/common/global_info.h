// stand-in for <hydra/common/global_info.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../../ref_standin.h"
