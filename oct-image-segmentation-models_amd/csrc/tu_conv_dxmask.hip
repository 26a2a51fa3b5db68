// One translation unit of liboct_unet_hip.so (see host.hpp): the MFMA conv launcher for <KH, addressing mode, epilogue> = <3, oct::A_NORMAL, oct::EPI_MASK>.
#include "launch_conv.hpp"
namespace octh { template int launch_igemm<3, oct::A_NORMAL, oct::EPI_MASK>(const oct::IgemmArgs&, const LaunchCtx&, int*); }
