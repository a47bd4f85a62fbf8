// stand-in for <hydra/common/graph_update.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../../ref_standin.h"
