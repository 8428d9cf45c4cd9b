// TEST INFRASTRUCTURE ONLY: the product kernels compiled against the SIMT
// emulator (see hip_emu.h).  Exposes the same C ABI as libblance_hip.so.
#include "hip_emu.h"
namespace emu {
Block* t_block = nullptr;
dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
}
#include "../../blance_amd/csrc/blance_hip.hip"
