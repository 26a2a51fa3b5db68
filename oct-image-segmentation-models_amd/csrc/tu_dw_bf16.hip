// One translation unit of liboct_unet_hip.so (see host.hpp): backward-weights kernels on the bf16 MFMA pipe.
#define OCT_TU_DW_BF16 1
#include "kernels_bx.hpp"
#include "launch_dw.hpp"
