// stand-in for <spark_dsg/node_symbol.h>: see ref_standin.h (oracle/ref_recipe/standin; test infrastructure)
#pragma once
#include "../ref_standin.h"
