// stand-in for <hydra/reconstruction/mesh_integrator.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../../ref_standin.h"
