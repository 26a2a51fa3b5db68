// One translation unit of liboct_unet_hip.so (see host.hpp): backward-weights kernels on the fp32 MFMA pipe.
#define OCT_TU_DW_F32 1
#include "kernels_dw.hpp"
#include "launch_dw.hpp"
