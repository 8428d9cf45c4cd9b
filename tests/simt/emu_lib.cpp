// TEST INFRASTRUCTURE ONLY: the product kernels compiled against the SIMT
// emulator (see hip_emu.h).  Exposes the same C ABI as libblance_hip.so.
#include "hip_emu.h"
#if !defined(__x86_64__)
#error "the SIMT emulator's context switch is written for x86-64"
#endif
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size emu_switch, .-emu_switch
)");
namespace emu {
Block* t_block = nullptr;
dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
std::mutex g_launch_mu;
}
#include "../../blance_amd/csrc/blance_hip.hip"
