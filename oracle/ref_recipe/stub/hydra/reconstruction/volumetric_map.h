// stand-in for the upstream header of the same name: see ../ref_api_stub.h (compile check of dump_vectors.cpp only)
#pragma once
#include "ref_api_stub.h"
