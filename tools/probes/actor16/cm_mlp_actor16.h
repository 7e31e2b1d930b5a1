// cm_mlp_actor16.h -- interface of the wave-private PPO actor kernel (cm_mlp_actor16.hip, its own translation unit and compiler flags)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

struct A16Args {  // only what this kernel reads (MlpArgs by value costs ~100 SGPRs)
    const float* x; const float* params; const uint8_t* avail; const int* action; const float* logp_old; const float* adv; const int* ep_len;
    float* partial;
    long rows, x_stride, avail_stride;
    int din, H, dout, A, T, PS;
    float clip_lo, clip_hi, clip_eps, ent_coef;
};

bool cm_actor16_enabled();                                    // CM_ACTOR_KERNEL=wave16
bool cm_actor16_supports(int din, int H, int L, int dout, bool rows_16B_aligned, int PS);
int cm_actor16_launch(const A16Args& a, hipStream_t s);      // returns the grid size (= number of partial-gradient rows written)
