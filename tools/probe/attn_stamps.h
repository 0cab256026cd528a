// attn_stamps.h — the timing hooks of attn32_bf16_kernel, defined by the PROBES (attn_bench.hip, attn_context.hip built with
// -DATTN32_STAMPS) before they include ../../image_search_amd/csrc/attn32_kernels.h.  The library leaves the three hook
// macros empty: its kernel carries no diagnostic code.  Per workgroup and wave, slot j accumulates the shader cycles
// between consecutive ATTN32_STAMP points (0 own loads landed, 1 barrier, 2 issue next, 3 whole tile, 4 split tile).
#pragma once
#include <hip/hip_runtime.h>
__device__ unsigned long long* attn32_stamp_buf;
#define ATTN32_STAMP_BEGIN unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#define ATTN32_STAMP(SLOT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (lane == 0) acc_[SLOT] += now_ - last_; last_ = __builtin_amdgcn_s_memtime(); }
#define ATTN32_STAMP_END if (lane == 0) for (int j = 0; j < 5; ++j) attn32_stamp_buf[((size_t)blockIdx.x * 8 + wave) * 8 + j] = acc_[j];
