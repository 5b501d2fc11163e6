// Register-level FFT butterflies for the in-LDS transforms (fft_lds.hpp): r-point DFTs, r = 2, 4, 8, 16, on values a thread holds
// in registers, built from radix-4 / radix-2 butterflies with the constant twiddles of the 8- and 16-point transforms.
//
// Single precision works on (re, im) register PAIRS with the packed-fp32 instructions of CDNA (v_pk_add_f32, v_pk_mul_f32,
// v_pk_fma_f32): a complex add is one instruction, a complex multiply two, and multiplication by +-i is free -- it is folded into
// the following add through the operand-select / negate modifiers (op_sel, op_sel_hi, neg_lo, neg_hi).  hipcc does not find
// these forms from scalar code (it packs the real parts of two different complex numbers into one register pair and shuffles with
// v_mov / v_pk_mov; measured on the round-1 kernels: 556 of the 3447 instructions of the delta-flow column kernel were moves), so
// the handful of complex primitives is written as single-instruction inline asm; everything above them is plain C++ and is
// scheduled by the compiler.  Double precision uses the same butterflies on cx<double> with scalar arithmetic.
//
// A radix-16 butterfly is 64 complex adds + 8 constant + 15 external twiddle multiplies = 110 packed instructions (the level-by-
// level radix-2 form it replaces: 64 adds + 42 multiplies + the moves).
#pragma once
#include "common.hpp"

namespace cmbl {

typedef float f2 __attribute__((ext_vector_type(2)));

// value type held in registers inside a butterfly
template <typename T> struct vreg { using type = cx<T>; };
template <> struct vreg<float> { using type = f2; };

// ---- single precision: packed primitives ---------------------------------------------------------------------------------
__device__ __forceinline__ f2 vadd(f2 a, f2 b) { return a + b; }
__device__ __forceinline__ f2 vsub(f2 a, f2 b) { return a - b; }
// a + i b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 vaddi(f2 a, f2 b) {
  f2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// a - i b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2 vsubi(f2 a, f2 b) {
  f2 r; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
// a * w = (a.x w.x - a.y w.y, a.x w.y + a.y w.x)
__device__ __forceinline__ f2 vmul(f2 a, f2 w) {
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                                      // (a.y w.y, a.y w.x)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));       // (a.x w.x - t.x, a.x w.y + t.y)
  return r;
}
// a * conj(w) = (a.x w.x + a.y w.y, a.y w.x - a.x w.y)
__device__ __forceinline__ f2 vmulc(f2 a, f2 w) {
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));                                      // (a.y w.y, a.x w.y)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));       // (a.x w.x + t.x, a.y w.x - t.y)
  return r;
}
// the same with a wave-uniform constant w held in a scalar register pair (no vector registers, no moves)
__device__ __forceinline__ f2 vmul_k(f2 a, f2 w) {
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "s"(w), "v"(t));
  return r;
}
// -i a = (a.y, -a.x) ;  i a = (-a.y, a.x)
__device__ __forceinline__ f2 vmul_mi(f2 a) { return f2{a.y, -a.x}; }
__device__ __forceinline__ f2 vmul_pi(f2 a) { return f2{-a.y, a.x}; }
__device__ __forceinline__ f2 vscale(f2 a, float s) { return a * s; }
__device__ __forceinline__ f2 vmake(float x, float y) { return f2{x, y}; }
__device__ __forceinline__ f2 vload(const cx<float>* p) { return *reinterpret_cast<const f2*>(p); }
__device__ __forceinline__ void vstore(cx<float>* p, f2 v) { *reinterpret_cast<f2*>(p) = v; }
__device__ __forceinline__ cx<float> vcx(f2 v) { return mk<float>(v.x, v.y); }
__device__ __forceinline__ f2 vfrom(cx<float> c) { return f2{c.x, c.y}; }

// ---- double precision: the same primitives, scalar -------------------------------------------------------------------------
__device__ __forceinline__ cx<double> vadd(cx<double> a, cx<double> b) { return a + b; }
__device__ __forceinline__ cx<double> vsub(cx<double> a, cx<double> b) { return a - b; }
__device__ __forceinline__ cx<double> vaddi(cx<double> a, cx<double> b) { return mk<double>(a.x - b.y, a.y + b.x); }
__device__ __forceinline__ cx<double> vsubi(cx<double> a, cx<double> b) { return mk<double>(a.x + b.y, a.y - b.x); }
__device__ __forceinline__ cx<double> vmul(cx<double> a, cx<double> w) { return a * w; }
__device__ __forceinline__ cx<double> vmulc(cx<double> a, cx<double> w) { return cmulconj(a, w); }
__device__ __forceinline__ cx<double> vmul_k(cx<double> a, cx<double> w) { return a * w; }
__device__ __forceinline__ cx<double> vmul_mi(cx<double> a) { return mul_mi(a); }
__device__ __forceinline__ cx<double> vmul_pi(cx<double> a) { return mul_i(a); }
__device__ __forceinline__ cx<double> vscale(cx<double> a, double s) { return s * a; }
__device__ __forceinline__ cx<double> vmake(double x, double y) { return mk<double>(x, y); }
__device__ __forceinline__ cx<double> vload(const cx<double>* p) { return *p; }
__device__ __forceinline__ void vstore(cx<double>* p, cx<double> v) { *p = v; }
__device__ __forceinline__ cx<double> vcx(cx<double> v) { return v; }
__device__ __forceinline__ cx<double> vfrom(cx<double> c) { return c; }

// ---- butterflies -------------------------------------------------------------------------------------------------------------
// exp(-+ 2 pi i e / 16) as a value of the register type (INV: conjugate)
template <typename T, bool INV> __device__ __forceinline__ typename vreg<T>::type w16(int e) {
  // cos / sin of e pi / 8 from the first octant, as literals (a local table of doubles ended up in scratch memory)
  constexpr double c1 = 0.92387953251128674, c2 = 0.70710678118654752, c3 = 0.38268343236508977;
  double cs, sn;
  switch (e & 15) {
    case 0: cs = 1; sn = 0; break;            case 1: cs = c1; sn = c3; break;     case 2: cs = c2; sn = c2; break;      case 3: cs = c3; sn = c1; break;
    case 4: cs = 0; sn = 1; break;            case 5: cs = -c3; sn = c1; break;    case 6: cs = -c2; sn = c2; break;     case 7: cs = -c1; sn = c3; break;
    case 8: cs = -1; sn = 0; break;           case 9: cs = -c1; sn = -c3; break;   case 10: cs = -c2; sn = -c2; break;   case 11: cs = -c3; sn = -c1; break;
    case 12: cs = 0; sn = -1; break;          case 13: cs = c3; sn = -c1; break;   case 14: cs = c2; sn = -c2; break;    default: cs = c1; sn = -c3; break;
  }
  return vmake((T)cs, (T)(INV ? sn : -sn));
}

// 4-point DFT in place, natural order in and out:  x_q <- sum_p x_p (-+i)^(p q)
template <bool INV, typename V> __device__ __forceinline__ void radix4(V& x0, V& x1, V& x2, V& x3) {
  const V t0 = vadd(x0, x2), t1 = vsub(x0, x2), t2 = vadd(x1, x3), t3 = vsub(x1, x3);
  x0 = vadd(t0, t2); x2 = vsub(t0, t2);
  x1 = INV ? vaddi(t1, t3) : vsubi(t1, t3);
  x3 = INV ? vsubi(t1, t3) : vaddi(t1, t3);
}

// the same with x2 standing for (-+i) x2 (forward: -i, inverse: +i): the multiplication rides in the first add / subtract
template <bool INV, typename V> __device__ __forceinline__ void radix4_rot2(V& x0, V& x1, V& x2, V& x3) {
  const V t0 = INV ? vaddi(x0, x2) : vsubi(x0, x2), t1 = INV ? vsubi(x0, x2) : vaddi(x0, x2), t2 = vadd(x1, x3), t3 = vsub(x1, x3);
  x0 = vadd(t0, t2); x2 = vsub(t0, t2);
  x1 = INV ? vaddi(t1, t3) : vsubi(t1, t3);
  x3 = INV ? vsubi(t1, t3) : vaddi(t1, t3);
}

// r-point DFT of v[0..r-1] (natural order), r = 2^LG.  Output X_k is left at v[dft_loc<LG>(k)].
template <int LG> __device__ __host__ constexpr int dft_loc(int k) {
  return LG == 4 ? 4 * (k & 3) + (k >> 2) : LG == 3 ? 4 * (k & 1) + (k >> 1) : k;
}
template <typename T, int LG, bool INV, typename V> __device__ __forceinline__ void dft(V (&v)[1 << LG]) {
  if constexpr (LG == 1) {
    const V a = v[0], b = v[1];
    v[0] = vadd(a, b); v[1] = vsub(a, b);
  } else if constexpr (LG == 2) {
    radix4<INV>(v[0], v[1], v[2], v[3]);
  } else if constexpr (LG == 3) {
    // radix-2 step (m, m+4) with W8^m on the difference, then two 4-point transforms:  X[2r] from the sums, X[2r+1] from the rest
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const V a = v[m], b = v[m + 4];
      v[m] = vadd(a, b);
      const V d = vsub(a, b);
      v[m + 4] = (m == 0 || m == 2) ? d : vmul_k(d, w16<T, INV>(2 * m));        // m = 2: the factor -+i is applied by radix4_rot2
    }
    radix4<INV>(v[0], v[1], v[2], v[3]);
    radix4_rot2<INV>(v[4], v[5], v[6], v[7]);
  } else {
    static_assert(LG == 4, "radix up to 16");
    // radix-4 over (m0, m0+4, m0+8, m0+12): b[q][m0] at v[m0 + 4q]; twiddle W16^(m0 q); radix-4 over m0: X[q + 4r] at v[4q + r]
#pragma unroll
    for (int m0 = 0; m0 < 4; ++m0) radix4<INV>(v[m0], v[m0 + 4], v[m0 + 8], v[m0 + 12]);
#pragma unroll
    for (int q = 1; q < 4; ++q)
#pragma unroll
      for (int m0 = 1; m0 < 4; ++m0) {
        const int e = m0 * q;
        V& x = v[m0 + 4 * q];
        if (e != 4) x = vmul_k(x, w16<T, INV>(e));                                // e = 4 (m0 = q = 2): -+i, applied by radix4_rot2
      }
    radix4<INV>(v[0], v[1], v[2], v[3]);
    radix4<INV>(v[4], v[5], v[6], v[7]);
    radix4_rot2<INV>(v[8], v[9], v[10], v[11]);
    radix4<INV>(v[12], v[13], v[14], v[15]);
  }
}

}  // namespace cmbl
