// MFX_SGD_TILED kernel for the rank shape L=16 lanes x C=7 chunks (see sgd_slots_kernel.h)
#include "sgd_slots_kernel.h"

MFX_SLOTS_INSTANCE(16, 7)
