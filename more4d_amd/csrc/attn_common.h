// Helpers shared by the attention forward and backward kernels: XOR-swizzled LDS tile addressing, the
// "swap index bits 2,3" row permutation, fragment reads and accumulator -> operand packing.
#pragma once
#include "common.h"

namespace {

template <int RB> M4D_DEV int swz_off(int row, int chunk) {
    constexpr int CPR = RB / 16;
    if constexpr (CPR >= 16) return row * RB + ((chunk ^ (row & 15)) << 4);
    else return row * RB + ((chunk ^ ((row / (16 / CPR)) & (CPR - 1))) << 4);
}

M4D_DEV int perm23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <typename T> struct TileCfg;
template <> struct TileCfg<bf16_t> { static constexpr int KVB = 64; };
template <> struct TileCfg<float> { static constexpr int KVB = 32; };

template <typename T> M4D_DEV typename Frag8<T>::type lds_frag(const char* base, int off0, int off1);
template <> M4D_DEV bf16x8 lds_frag<bf16_t>(const char* base, int off0, int) {
    return *reinterpret_cast<const bf16x8*>(base + off0);
}
template <> M4D_DEV f32x8 lds_frag<float>(const char* base, int off0, int off1) {
    f32x4 lo = *reinterpret_cast<const f32x4*>(base + off0);
    f32x4 hi = *reinterpret_cast<const f32x4*>(base + off1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <typename T> M4D_DEV typename Frag8<T>::type pack8(const f32x16& s, int base);
template <> M4D_DEV bf16x8 pack8<bf16_t>(const f32x16& s, int base) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)s[base + j];
    return r;
}
template <> M4D_DEV f32x8 pack8<float>(const f32x16& s, int base) {
    f32x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = s[base + j];
    return r;
}

}  // namespace
