// one Poseidon1-16 permutation per lane, straight line: the instruction mix behind the sponge ceiling (tools/isa_mix.py --all -> "poseidon16_permute_one_lane")
#include <hip/hip_runtime.h>
#include "../../leanmultisig_amd/csrc/poseidon16.h"
using namespace kb;
__global__ __launch_bounds__(256) void k_one_perm(u32* io) {
    u32 s[16];
    const u32 t = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = io[t * 16 + i];
    poseidon16_permute(s);
#pragma unroll
    for (int i = 0; i < 16; i++) io[t * 16 + i] = s[i];
}
