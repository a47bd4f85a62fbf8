// stand-in for <config_utilities/virtual_config.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../ref_standin.h"
