// MFX_SGD_TILED kernel for the rank shape L=4 lanes x C=1 chunks (see sgd_slots_kernel.h)
#include "sgd_slots_kernel.h"

MFX_SLOTS_INSTANCE(4, 1)
