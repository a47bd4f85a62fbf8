// stand-in for <hydra/openset/embedding_group.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../../ref_standin.h"
