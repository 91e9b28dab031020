// vg_target.h — everything that is spelled differently for gfx950 and for the CPU fiber emulation of tests/simt (which compiles
// these same kernel sources with g++: one address space, no DPP / readlane hardware, no kernarg segment), in ONE place, so that
// the kernel translation units themselves carry no emulator branches.  Each item: the gfx950 form and why it is written that way,
// then the plain form the emulator uses.  Nothing here is a second compute path: the emulated form is the same arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef VINS_SIMT
// ---- address spaces.  A pointer handed to a non-inlined function, or loaded from a table in HBM, is generic, and generic
//      accesses compile to flat_load / flat_store even when they always hit LDS or always hit HBM (slower issue and latency, and
//      they tie up both memory counters): the kernels re-type such operands explicitly.
typedef __attribute__((address_space(3))) double lds_d;
typedef __attribute__((address_space(3))) int lds_i;
typedef __attribute__((address_space(1))) double glb_d;
typedef __attribute__((address_space(1))) int glb_i;
typedef __attribute__((address_space(1))) uint8_t glb_u8;
typedef __attribute__((address_space(1))) uint32_t glb_u32;
typedef __attribute__((address_space(1))) uint16_t glb_u16;
typedef uint32_t vg_u32_a2 __attribute__((aligned(2)));
typedef __attribute__((address_space(1))) const vg_u32_a2 glb_u32_a2;      // a dword in HBM at a 2-byte aligned address: one global_load_dword
#define VG_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n, n)))
// ---- the kernarg segment (explicit kernel arguments, then the hidden ones): constant address space
typedef const __attribute__((address_space(4))) char* vg_kernarg_ptr;
// ---- the constant-rate wall clock behind wall_clock64(): 100 MHz on gfx950
#define BA_WALL_HZ 1e8
// ---- a pointer the optimiser must consider used (keeps a call out of tail position: see PHASE_ENTER in ba_pipeline.hip)
#define VG_KEEP_ALIVE(p) asm volatile("" :: "v"(p))
// ---- exact product of two integers known to fit 24 bits whose product fits 32: v_mul_i32_i24 runs at full rate, v_mul_lo_u32 at
//      a quarter of it
__device__ __forceinline__ int vg_mul24(int a, int b) { return __mul24(a, b); }
// ---- uni(): a value every lane of the wavefront holds identically (solver control state, results of workgroup reductions), marked
//      as such (v_readfirstlane): the compiler keeps it in SGPRs.  What is live in VGPRs across a call to a non-inlined phase
//      function is saved to scratch per lane; uniform state in SGPRs is not.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// ---- const_load(): a block the host wrote before the launch and no kernel ever writes (the batch layout), read through the
//      constant address space: every field becomes a scalar load (s_load) and everything derived from it -- strides, buffer
//      offsets, the pointers of a window -- wave-uniform SGPR arithmetic
template <typename T>
__device__ __forceinline__ T const_load(const T* p) {
    T v;
    __builtin_memcpy(&v, (const __attribute__((address_space(4))) void*)p, sizeof(T));
    return v;
}
// ---- 16 bytes from an arbitrary byte address of an image plane in HBM: one global_load_dwordx4
__device__ __forceinline__ uint4 vg_load16_unaligned(const glb_u8* p) {
    typedef unsigned int nv4 __attribute__((ext_vector_type(4), aligned(1)));
    typedef __attribute__((address_space(1))) const nv4 glb_nv4;
    const nv4 t4 = *(glb_nv4*)p;
    uint4 v;
    v.x = t4.x; v.y = t4.y; v.z = t4.z; v.w = t4.w;
    return v;
}
// ---- four ints from a 16-byte aligned table entry in HBM: one global_load_dwordx4
__device__ __forceinline__ int4 vg_load_int4(const void* p) {
    typedef int nv4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const nv4 glb_nv4;
    const nv4 t4 = *(glb_nv4*)p;
    int4 v;
    v.x = t4.x; v.y = t4.y; v.z = t4.z; v.w = t4.w;
    return v;
}
// ---- bytes q and q + 1 of the eight bytes (hi:lo) as two 16-bit lanes (byte q | byte q + 1 << 16): one v_perm_b32
template <int Q> __device__ __forceinline__ unsigned vg_byte_pair(unsigned hi, unsigned lo) {
    return __builtin_amdgcn_perm(hi, lo, (unsigned)Q | (0x0Cu << 8) | ((unsigned)(Q + 1) << 16) | (0x0Cu << 24));
}

// ---- v_perm_b32 with a constant selector (result byte i = byte SEL.i of the eight bytes hi:lo, 0x0C = zero), and arithmetic on
//      two unsigned 16-bit lanes (v_pk_add_u16 / v_pk_mad_u16 / v_pk_lshrrev_b16: wrap-around per lane, the callers stay below 2^16)
template <unsigned SEL> __device__ __forceinline__ unsigned vg_perm(unsigned hi, unsigned lo) { return __builtin_amdgcn_perm(hi, lo, SEL); }
typedef unsigned short vg_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned vg_pk_add(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(vg_us2, a) + __builtin_bit_cast(vg_us2, b)); }
__device__ __forceinline__ unsigned vg_pk_sub(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(vg_us2, a) - __builtin_bit_cast(vg_us2, b)); }
__device__ __forceinline__ unsigned vg_pk_mad(unsigned a, unsigned short k, unsigned c) {
    const vg_us2 kk = {k, k};
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(vg_us2, a) * kk + __builtin_bit_cast(vg_us2, c));
}
__device__ __forceinline__ unsigned vg_pk_shr(unsigned a, unsigned short n) {
    const vg_us2 nn = {n, n};
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(vg_us2, a) >> nn);
}

// ---- signed dot product of two pairs of int16 lanes plus a 32-bit addend (v_dot2_i32_i16)
__device__ __forceinline__ int vg_sdot2(unsigned a, unsigned b, int c) {
    typedef short s2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, a), __builtin_bit_cast(s2, b), c, false);
}
// ---- the same with the addend in a register that stays live (a per-pixel constant reused by every iteration): the compiler
//      selects the two-operand v_dot2c_i32_i16 (destination = addend) behind a v_mov copy of the addend; the three-operand VOP3P
//      form needs no copy
__device__ __forceinline__ int vg_sdot2_keep(unsigned a, unsigned b, int c) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#else   // ------------------------------------------------------------------------------ CPU fiber emulation (tests/simt)
typedef double lds_d;
typedef int lds_i;
typedef double glb_d;
typedef int glb_i;
typedef uint8_t glb_u8;
typedef uint32_t glb_u32;
typedef uint16_t glb_u16;
typedef uint32_t vg_u32_a2 __attribute__((aligned(2)));
typedef const vg_u32_a2 glb_u32_a2;
#define VG_WAVES_PER_EU(n)
typedef const char* vg_kernarg_ptr;                        // (hipLaunchKernelGGL of the emulator packs the arguments the same way)
#define BA_WALL_HZ 1e9                                     // (the emulated clock counts nanoseconds)
#define VG_KEEP_ALIVE(p) ((void)(p))
inline int vg_mul24(int a, int b) { return a * b; }
inline int uni(int v) { return v; }
inline double uni(double v) { return v; }
template <typename T> inline T const_load(const T* p) { return *p; }
inline uint4 vg_load16_unaligned(const glb_u8* p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
inline int4 vg_load_int4(const void* p) { int4 v; __builtin_memcpy(&v, p, 16); return v; }
template <int Q> inline unsigned vg_byte_pair(unsigned hi, unsigned lo) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    return (unsigned)((v >> (8 * Q)) & 255u) | ((unsigned)((v >> (8 * (Q + 1))) & 255u) << 16);
}
template <unsigned SEL> inline unsigned vg_perm(unsigned hi, unsigned lo) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned sel = (SEL >> (8 * i)) & 255u;
        if (sel < 8) r |= (unsigned)((v >> (8 * sel)) & 255u) << (8 * i);
        else if (sel != 0x0Cu) { fprintf(stderr, "simt: unmodelled v_perm_b32 selector 0x%x\n", sel); abort(); }
    }
    return r;
}
inline unsigned vg_pk_add(unsigned a, unsigned b) { return ((a + b) & 0xffffu) | (((a >> 16) + (b >> 16)) << 16); }
inline unsigned vg_pk_sub(unsigned a, unsigned b) { return ((a - b) & 0xffffu) | (((a >> 16) - (b >> 16)) << 16); }
inline unsigned vg_pk_mad(unsigned a, unsigned short k, unsigned c) { return (((a & 0xffffu) * k + (c & 0xffffu)) & 0xffffu) | (((a >> 16) * k + (c >> 16)) << 16); }
inline unsigned vg_pk_shr(unsigned a, unsigned short n) { return ((a & 0xffffu) >> n) | (((a >> 16) >> n) << 16); }
inline int vg_sdot2(unsigned a, unsigned b, int c) { return (int)(short)(a & 0xffffu) * (int)(short)(b & 0xffffu) + (int)(short)(a >> 16) * (int)(short)(b >> 16) + c; }
inline int vg_sdot2_keep(unsigned a, unsigned b, int c) { return vg_sdot2(a, b, c); }
#endif
