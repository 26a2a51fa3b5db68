// One translation unit of liboct_unet_hip.so (see host.hpp): the MFMA conv launcher for <KH, addressing mode, epilogue> = <2, oct::A_UPF, oct::EPI_FWD>.
#include "launch_conv.hpp"
namespace octh { template int launch_igemm<2, oct::A_UPF, oct::EPI_FWD>(const oct::IgemmArgs&, const LaunchCtx&, int*); }
