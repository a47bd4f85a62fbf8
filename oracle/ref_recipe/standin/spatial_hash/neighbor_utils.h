// stand-in for <spatial_hash/neighbor_utils.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../ref_standin.h"
