// Loop bodies of the reference's drivers as library calls, for hosts that are neither Julia (which keeps src/maximization.jl:116-233
// and src/sampling.jl:388-464 itself) nor Python (cmblensing.jl_amd/drivers.py holds the same control flow):
//   hmc_step         src/sampling.jl:405-418 over symplectic_integrate (:14-46) with U = logpdf(Mixed(ds)) as a function of phi°
//   map_joint_step   src/maximization.jl:160-206: Wiener filter f | phi, d logpdf(Mixed) / d phi°, step direction, Brent line search, unmix
// Control flow on the host, every field operation one of the engine's launches; all fields in the internal F layout here (the C ABI
// wrappers in api.hip convert at the edges).  Nothing new is computed on the device except two plane-wise helpers (sqrt, pinv).
#pragma once
#include "engine.hpp"

namespace cmbl {

// out = sqrt(in) (mode 0) or pinv(in) = 1 / in with non-finite -> 0 (mode 1), real planes
template <typename T>
__global__ __launch_bounds__(NTP) void k_plane_unary(const T* __restrict__ in, T* __restrict__ out, long n, int mode) {
  const long i = (long)blockIdx.x * NTP + threadIdx.x;
  if (i >= n) return;
  const T v = in[i];
  if (mode == 0) out[i] = sqrt(v);
  else { const T r = T(1) / v; out[i] = isfinite(r) ? r : T(0); }
}

// Brent's bounded 1-D minimiser (golden section + successive parabolic interpolation): the algorithm behind
// `Optim.optimize(f, lo, hi, Brent(); abs_tol)` at src/maximization.jl:194-199 (the iterate sequence of Optim.jl itself is unpinned,
// DESIGN.md §3; only the converged alpha matters).  Same code as drivers.py brent_minimize.  Returns x; *fmin, *nfev optional.
template <typename F>
inline double brent_minimize(F&& f, double lo, double hi, double abs_tol, double rel_tol, double* fmin, int* nfev, int max_iter = 1000) {
  const double golden = (3 - std::sqrt(5.0)) / 2;
  double x = lo + golden * (hi - lo), w = x, v = x;
  double fx = f(x), fw = fx, fv = fx;
  int n = 1;
  double step = 0, old_step = 0;
  for (int it = 0; it < max_iter; ++it) {
    const double mid = (lo + hi) / 2, tol = rel_tol * std::fabs(x) + abs_tol;
    if (std::fabs(x - mid) <= 2 * tol - (hi - lo) / 2) break;
    double p = 0, q = 0;
    if (std::fabs(old_step) > tol) {
      const double r = (x - w) * (fx - fv);
      q = (x - v) * (fx - fw);
      p = (x - v) * q - (x - w) * r;
      q = 2 * (q - r);
      if (q > 0) p = -p; else q = -q;
    }
    if (std::fabs(p) < std::fabs(q * old_step / 2) && p > q * (lo - x) && p < q * (hi - x)) {
      old_step = step; step = p / q;
      const double xt = x + step;
      if ((xt - lo) < 2 * tol || (hi - xt) < 2 * tol) step = x < mid ? tol : -tol;
    } else {
      old_step = x < mid ? hi - x : lo - x;
      step = golden * old_step;
    }
    const double u = x + (std::fabs(step) >= tol ? step : (step > 0 ? tol : -tol));
    const double fu = f(u);
    ++n;
    if (fu <= fx) {
      if (u < x) hi = x; else lo = x;
      v = w; fv = fw; w = x; fw = fx; x = u; fx = fu;
    } else {
      if (u < x) lo = u; else hi = u;
      if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; }
      else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
    }
  }
  if (fmin) *fmin = fx;
  if (nfev) *nfev = n;
  return x;
}

// counter-based uniforms of the drivers (cmblensing.jl_amd/rng.py `uniform`): word j of Philox4x32-10 at counter (c, stream), key = seed
inline void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
inline double philox_uniform(uint64_t seed, uint64_t stream) {
  uint32_t c[4] = {0, 0, (uint32_t)stream, (uint32_t)(stream >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (c[0] + 0.5) / 4294967296.0;
}
constexpr int STREAM_P = 2, STREAM_U = 3;                                  // rng.py: draw kind + 16 * step
inline uint64_t stream_id(int kind, uint64_t step) { return (uint64_t)kind + 16 * step; }

template <typename T>
struct Drivers {
  Dataset<T>& ds;
  Flow<T>& L;
  Ctx<T>* c;
  DevBuf planes, xa, xb, pa, ga, gb, ta, gmap, fo_buf, fh_buf;
  std::vector<double> one, lp;

  Drivers(Dataset<T>& d, Flow<T>& l) : ds(d), L(l), c(d.c) {}

  void plane_unary(const T* in, T* out, int mode) {
    const long n = c->plane();
    CMBL_LAUNCH(c, K_LINCOMB, (k_plane_unary<T>), dim3((unsigned)((n + NTP - 1) / NTP)), 0, c->stream, in, out, n, mode);
  }
  // out = plane .* in  (S0 Fourier fields, F layout)
  void mul_plane(const T* plane, const cx<T>* in, cx<T>* out, int B) {
    const T* d[5] = {plane, nullptr, nullptr, nullptr, nullptr};
    c->harm(in, out, 1, B, 1, d, false, false, false);
  }
  void axpby(cx<T>* out, const double* a, const cx<T>* x, const double* b, const cx<T>* y, int B) {
    c->lincomb((T*)out, (const T*)x, (const T*)y, a, b, 2 * c->plane(), B);
  }
  void axpby1(cx<T>* out, double a, const cx<T>* x, double b, const cx<T>* y, int B) {
    std::vector<double> va(B, a), vb(B, b);
    axpby(out, va.data(), x, vb.data(), y, B);
  }

  // ---- hmc_step --------------------------------------------------------------------------------------------------------------------
  // fo: map; phio_F, out_F: S0 Fourier F layout; mass_F: real plane F layout (Lambda = mass_matrix_phi, src/sampling.jl:422-425);
  // white_p: unit white-noise map (1, B) -- the momentum is p0 = sqrt(Lambda) * rfft(white_p) (:407, simulate(Diagonal))
  void hmc_step(const T* fo, const cx<T>* phio_F, const T* mass_F, const T* white_p, const double* log_u, int nleap, double eps, bool always_accept,
                bool quirk, int B, cx<T>* out_F, double* dH, int* accept) {
    const long pl = c->plane(), sl = (long)ds.P * B;
    planes.ensure(sizeof(T) * 2 * pl);
    T* sq = planes.as<T>(); T* li = sq + pl;
    plane_unary(mass_F, sq, 0); plane_unary(mass_F, li, 1);
    xa.ensure(sizeof(cx<T>) * B * pl); xb.ensure(sizeof(cx<T>) * B * pl); pa.ensure(sizeof(cx<T>) * B * pl);
    ga.ensure(sizeof(cx<T>) * B * pl); gb.ensure(sizeof(cx<T>) * B * pl); ta.ensure(sizeof(cx<T>) * B * pl);
    gmap.ensure(sizeof(T) * sl * c->npix());
    cx<T>*x = xa.as<cx<T>>(), *x1 = xb.as<cx<T>>(), *p = pa.as<cx<T>>(), *g = ga.as<cx<T>>(), *g1 = gb.as<cx<T>>(), *t = ta.as<cx<T>>();
    lp.assign(B, 0.0);
    std::vector<double> kin0(B), kin1(B), u0(B), u1(B);
    // p0 = sqrt(Lambda) * rfft(white)
    c->rfft2_F(white_p, p, B);
    mul_plane(sq, p, p, B);
    CMBL_HIP(hipMemcpyAsync(x, phio_F, sizeof(cx<T>) * B * pl, hipMemcpyDeviceToDevice, c->stream));
    auto U = [&](const cx<T>* xx, double* out) { ds.logpdf_mixed(L, fo, xx, out, nullptr, nullptr, B, quirk); };
    auto dU = [&](const cx<T>* xx, cx<T>* gg) { ds.logpdf_mixed(L, fo, xx, lp.data(), gmap.as<T>(), gg, B, quirk); };
    auto kinetic = [&](const cx<T>* pp, double* out) { mul_plane(li, pp, t, B); c->dot_F(pp, t, 1, B, out); };     // p' Lambda^-1 p
    U(x, u0.data()); kinetic(p, kin0.data());                            // H(x0, p0) = U - p'Λ⁻¹p / 2   (:21)
    dU(x, g);
    for (int i = 0; i < nleap; ++i) {                                    // :27-43
      axpby1(t, 1.0, p, -eps / 2, g, B);                                 // p - eps/2 g
      mul_plane(li, t, t, B);
      axpby1(x1, 1.0, x, -eps, t, B);                                    // x1 = x - eps Λ⁻¹ (p - eps/2 g)
      dU(x1, g1);
      axpby1(t, 1.0, g1, 1.0, g, B);
      axpby1(p, 1.0, p, -eps / 2, t, B);                                 // p = p - eps/2 (g1 + g)
      std::swap(x, x1); std::swap(g, g1);
    }
    U(x, u1.data()); kinetic(p, kin1.data());
    std::vector<double> acc(B), rej(B);
    for (int b = 0; b < B; ++b) {
      dH[b] = (u1[b] - kin1[b] / 2) - (u0[b] - kin0[b] / 2);
      const bool ok = always_accept || (log_u[b] < dH[b]);               // a NaN dH (diverged trajectory) compares false: rejected (:414)
      accept[b] = ok ? 1 : 0; acc[b] = ok ? 1.0 : 0.0; rej[b] = 1.0 - acc[b];
    }
    axpby(out_F, acc.data(), x, rej.data(), phio_F, B);                  // x = accept * xtest + (1 - accept) * x   (:415)
  }

  // ---- MAP_joint step ----------------------------------------------------------------------------------------------------------------
  // phi_F, phi_out_F: S0 Fourier F; fstart_h (may be nullptr), f_out_h: harmonic F; hinv_F: real plane F (pinv(Cphi^-1 + Nphi^-1),
  // src/dataset.jl:134-137).  The dataset's G must be the identity (the reference's MAP_joint sets ds.G = I, src/maximization.jl:146):
  // the wrapper swaps the operator in and out.
  void map_joint_step(const cx<T>* phi_F, const cx<T>* fstart_h, const T* hinv_F, double alpha_max, double alpha_tol, double cg_tol, int cg_maxit,
                      bool quirk, int B, cx<T>* f_out_h, cx<T>* phi_out_F, double* logpdf, double* alpha_out, int* ncg, int* nls, double* hist) {
    const long pl = c->plane(), sl = (long)ds.P * B, n = ds.fsize(B);
    L.set_phi_F(phi_F, B);
    *ncg = ds.wiener_cg(L, ds.d_h.template as<cx<T>>(), fstart_h, cg_tol, cg_maxit, f_out_h, hist, B);          // :164-169
    // mix (:176; src/dataset.jl:96-101): f° = L(phi) D f, phi° = G phi = phi
    fh_buf.ensure(sizeof(cx<T>) * n); fo_buf.ensure(sizeof(T) * sl * c->npix()); gmap.ensure(sizeof(T) * sl * c->npix());
    cx<T>* t = fh_buf.as<cx<T>>();
    ds.apply(OP_D, f_out_h, t, B, false, false, true);                   // harmonic -> D -> QU Fourier
    c->F_to_map(t, gmap.as<T>(), sl);
    L.flow_map(gmap.as<T>(), fo_buf.as<T>(), ds.P, B, false);
    const T* fo = fo_buf.as<T>();
    xa.ensure(sizeof(cx<T>) * B * pl); ga.ensure(sizeof(cx<T>) * B * pl); ta.ensure(sizeof(cx<T>) * B * pl);
    cx<T>*po = xa.as<cx<T>>(), *g = ga.as<cx<T>>(), *trial = ta.as<cx<T>>();
    CMBL_HIP(hipMemcpyAsync(po, phi_F, sizeof(cx<T>) * B * pl, hipMemcpyDeviceToDevice, c->stream));
    lp.assign(B, 0.0);
    ds.logpdf_mixed(L, fo, po, lp.data(), gmap.as<T>(), g, B, quirk);    // :178
    mul_plane(hinv_F, g, g, B);                                          // step direction (:188)
    const double big = (double)std::numeric_limits<T>::max();
    int nfev = 0;
    auto neg = [&](double a) {
      axpby1(trial, 1.0, po, a, g, B);
      ds.logpdf_mixed(L, fo, trial, lp.data(), nullptr, nullptr, B, quirk);
      double v = 0;
      for (int b = 0; b < B; ++b) v -= lp[b];
      return std::isnan(v) ? (a / alpha_max) * big : v;                  // :198
    };
    const double alpha = brent_minimize(neg, 0.0, alpha_max, alpha_tol, std::sqrt((double)std::numeric_limits<T>::epsilon()), nullptr, &nfev);   // :194-199
    axpby1(phi_out_F, 1.0, po, alpha, g, B);                             // :201; unmix of phi° with G = I is the identity (:206)
    ds.logpdf_mixed(L, fo, phi_out_F, logpdf, nullptr, nullptr, B, quirk);   // :205
    *alpha_out = alpha; *nls = nfev;
  }
};

}  // namespace cmbl

namespace cmbl {

// ---- quadratic_estimate (src/quadratic_estimate.jl:29-200) ---------------------------------------------------------------------------
// The Hu-Okamoto estimator with unlensed weights in the Fourier-diagonal approximation of the data model, exactly the sums of
// products of legs `QE_leg(C, inds...)` (:83-91) the reference writes down, memoised per (field, |l|-power, count of x / y indices)
// like its @memoize.  Same control flow as drivers.py quadratic_estimate (which the tests compare with the oracle).
// An index is a coordinate 1 / 2; `unit` marks a unit-vector index (l_i / |l|) as opposed to a derivative index (i l_i).
struct QEInd { int v; bool unit; };
inline QEInd D(int v) { return QEInd{v, false}; }     // derivative index  (the reference's tuple-wrapped `(i,)`)
inline QEInd Uv(int v) { return QEInd{v, true}; }     // unit-vector index

template <typename T>
struct QuadEst {
  Ctx<T>* c;
  int B;
  std::vector<std::unique_ptr<DevBuf>>& pool;                  // device buffers kept by the caller between calls (no hipMalloc after the first)
  size_t used = 0;
  std::map<std::tuple<const void*, int, int, int>, T*> legs;

  void* take(size_t bytes) {
    if (used == pool.size()) pool.emplace_back(new DevBuf());
    DevBuf& b = *pool[used++];
    b.ensure(bytes);
    return b.p;
  }
  cx<T>* new_fourier(int nb) { return (cx<T>*)take(sizeof(cx<T>) * nb * c->plane()); }
  T* new_map(int nb) { return (T*)take(sizeof(T) * nb * c->npix()); }

  // a real (Nyh x Nx, reference layout) double plane on the device: the caller's pointer itself when it is device memory, else a copy
  const double* dev_plane(const double* p, long n) {
    if (!p) return nullptr;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) == hipSuccess) { if (a.type == hipMemoryTypeDevice) return p; }
    else (void)hipGetLastError();                                  // plain host memory is "invalid value" to the runtime: not an error here
    double* d = (double*)take(sizeof(double) * n);
    CMBL_HIP(hipMemcpyAsync(d, p, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    return d;
  }
  const T* leg(const cx<T>* C, int nb, std::initializer_list<QEInd> inds) {
    int n = 0, p1 = 0, p2 = 0;
    for (const QEInd& q : inds) { n += q.unit ? 1 : 0; p1 += q.v == 1; p2 += q.v == 2; }
    auto key = std::make_tuple((const void*)C, n, p1, p2);
    auto it = legs.find(key);
    if (it == legs.end()) {
      T* m = new_map(nb);
      c->qe_leg(C, m, n, p1, p2, nb);
      it = legs.emplace(key, m).first;
    }
    return it->second;
  }
  // acc (+)= s * a .* b ; acc == nullptr allocates
  T* mul(const T* a, const T* b, double s, T* acc, int nb) {
    const bool fresh = acc == nullptr;
    if (fresh) acc = new_map(nb);
    c->map_fma(acc, a, b, s, !fresh, (long)nb * c->npix());
    return acc;
  }
};
inline int eps3(int a, int b) { return (a == 1 && b == 2) ? 1 : ((a == 2 && b == 1) ? -1 : 0); }

// planes (host, reference layout [x][ky], double): for each of the ncomp components (TT: T; EE: E; EB: E then B) Cf, Cftilde, Cn and
// TF = Mf .* B.  dref[comp]: that component of the data, S0 Fourier reference layout, B slots (device).
template <typename T>
void quadratic_estimate(Ctx<T>* c, std::vector<std::unique_ptr<DevBuf>>& pool, int which, int B, const cx<T>* const* dref, const double* Cf, const double* Cft, const double* Cn, const double* TF,
                        const double* Cphi, bool wiener, const double* AL_in, cx<T>* phiqe_ref, double* AL_out) {
  const long pl = c->plane();
  const int ncomp = which == 2 ? 2 : 1;
  QuadEst<T> q{c, B, pool};
  const unsigned gpl = (unsigned)std::min<long>((pl + NTP - 1) / NTP, 4096);
  // inverse-variance filter and the weight planes of orders 0, 1, 2 in Cf (:52-62, 100-110): one pointwise launch per component on the
  // device (round 4 formed them on the host: 15 planes of 2 M doubles, each uploaded behind a blocking synchronisation -- 227 ms at 2048^2
  // fp64, three times the Python driver)
  const double* dCf = q.dev_plane(Cf, ncomp * pl); const double* dCft = q.dev_plane(Cft, ncomp * pl);
  const double* dCn = q.dev_plane(Cn, ncomp * pl); const double* dTF = q.dev_plane(TF, ncomp * pl);
  const cx<T>*W0[2] = {nullptr, nullptr}, *W1[2] = {nullptr, nullptr}, *W2[2] = {nullptr, nullptr}, *fil[2] = {nullptr, nullptr}, *filC[2] = {nullptr, nullptr};
  for (int k = 0; k < ncomp; ++k) {
    cx<T>*w0 = q.new_fourier(1), *w1 = q.new_fourier(1), *w2 = q.new_fourier(1), *f = q.new_fourier(1), *fc = q.new_fourier(1);
    CMBL_LAUNCH(c, K_HARM, (k_qe_weights<T>), dim3(gpl), 0, c->stream, dCf + k * pl, dCft + k * pl, dCn + k * pl, dTF + k * pl, w0, w1, w2, f, fc, pl);
    W0[k] = w0; W1[k] = w1; W2[k] = w2; fil[k] = f; filC[k] = fc;
  }
  // filt: plane .* data component, in the reference layout (both operands are in it): the elementwise product of the interleaved reals
  auto filt = [&](int k, const cx<T>* w) {
    cx<T>* out = q.new_fourier(B);
    for (int b = 0; b < B; ++b) c->map_fma((T*)(out + (long)b * pl), (const T*)w, (const T*)(dref[k] + (long)b * pl), 1.0, false, 2 * pl);
    return (const cx<T>*)out;
  };
  cx<T>* un = q.new_fourier(B);
  T* tmp = nullptr;
  auto lmul = [&](const T* m, int p1, int p2, bool abs_, int nb) { cx<T>* o = q.new_fourier(nb); c->fourier_lmul(m, o, p1, p2, abs_, nb); return o; };
  auto add_to = [&](cx<T>* acc, double sa, const cx<T>* t, double st, bool first, int nb) {
    std::vector<double> a(nb, first ? 0.0 : sa), b(nb, st);
    c->lincomb((T*)acc, (const T*)(first ? t : acc), (const T*)t, a.data(), b.data(), 2 * pl, nb);
  };
  std::function<T*(int, int)> A;
  const int idx[2] = {1, 2};
  if (which == 0) {                                                                       // TT (:95-112)
    const cx<T>* a = filt(0, fil[0]); const cx<T>* b = filt(0, filC[0]);
    for (int i : idx) {
      tmp = q.mul(q.leg(a, B, {}), q.leg(b, B, {D(i)}), 1.0, nullptr, B);
      add_to(un, 1.0, lmul(tmp, i == 1, i == 2, false, B), -1.0, i == 1, B);
    }
    A = [&, W0, W1, W2](int i, int j) {
      T* acc = q.mul(q.leg(W2[0], 1, {D(i), D(j)}), q.leg(W0[0], 1, {}), 1.0, nullptr, 1);
      return q.mul(q.leg(W1[0], 1, {D(i)}), q.leg(W1[0], 1, {D(j)}), 1.0, acc, 1);
    };
  } else if (which == 1) {                                                                // EE (:115-140)
    const cx<T>* a1 = filt(0, filC[0]); const cx<T>* a2 = filt(0, fil[0]);
    for (int i : idx) {
      T* acc = nullptr;
      for (int j : idx) for (int k : idx) acc = q.mul(q.leg(a1, B, {D(i), Uv(j), Uv(k)}), q.leg(a2, B, {Uv(j), Uv(k)}), -2.0, acc, B);
      acc = q.mul(q.leg(a1, B, {D(i)}), q.leg(a2, B, {}), 1.0, acc, B);
      add_to(un, 1.0, lmul(acc, i == 1, i == 2, false, B), 1.0, i == 1, B);
    }
    A = [&, W0, W1, W2](int i, int j) {
      T* acc = nullptr;
      for (int k : idx) for (int l : idx) for (int m : idx) for (int n : idx) for (int p : idx) for (int qq : idx) {
        const int e = eps3(m, p) * eps3(n, qq);
        if (!e) continue;
        acc = q.mul(q.leg(W2[0], 1, {D(i), D(j), Uv(k), Uv(l), Uv(m), Uv(n)}), q.leg(W0[0], 1, {Uv(k), Uv(l), Uv(p), Uv(qq)}), -4.0 * e, acc, 1);
        acc = q.mul(q.leg(W1[0], 1, {D(i), Uv(k), Uv(l), Uv(m), Uv(n)}), q.leg(W1[0], 1, {D(j), Uv(k), Uv(l), Uv(p), Uv(qq)}), -4.0 * e, acc, 1);
      }
      acc = q.mul(q.leg(W2[0], 1, {D(i), D(j)}), q.leg(W0[0], 1, {}), 1.0, acc, 1);
      return q.mul(q.leg(W1[0], 1, {D(i)}), q.leg(W1[0], 1, {D(j)}), 1.0, acc, 1);
    };
  } else {                                                                                // EB (:143-175)
    const cx<T>*e1 = filt(0, fil[0]), *b2 = filt(1, fil[1]), *ce1 = filt(0, filC[0]), *cb2 = filt(1, filC[1]);
    for (int i : idx) {
      T* acc = nullptr;
      for (int j : idx) for (int k : idx) for (int l : idx) {
        const int e = eps3(k, l);
        if (!e) continue;
        acc = q.mul(q.leg(ce1, B, {D(i), Uv(j), Uv(k)}), q.leg(b2, B, {Uv(j), Uv(l)}), 2.0 * e, acc, B);
        acc = q.mul(q.leg(e1, B, {Uv(j), Uv(k)}), q.leg(cb2, B, {D(i), Uv(j), Uv(l)}), -2.0 * e, acc, B);
      }
      add_to(un, 1.0, lmul(acc, i == 1, i == 2, false, B), 1.0, i == 1, B);
    }
    A = [&, W0, W1, W2](int i, int j) {
      T* acc = nullptr;
      for (int k : idx) for (int l : idx) for (int m : idx) for (int n : idx) for (int p : idx) for (int qq : idx) {
        const int e = eps3(m, p) * eps3(n, qq);
        if (!e) continue;
        acc = q.mul(q.leg(W2[0], 1, {D(i), D(j), Uv(k), Uv(l), Uv(m), Uv(n)}), q.leg(W0[1], 1, {Uv(k), Uv(l), Uv(p), Uv(qq)}), 4.0 * e, acc, 1);
        acc = q.mul(q.leg(W1[0], 1, {D(i), Uv(k), Uv(l), Uv(m), Uv(n)}), q.leg(W1[1], 1, {D(j), Uv(k), Uv(l), Uv(p), Uv(qq)}), -8.0 * e, acc, 1);
        acc = q.mul(q.leg(W0[0], 1, {Uv(k), Uv(l), Uv(m), Uv(n)}), q.leg(W2[1], 1, {D(i), D(j), Uv(k), Uv(l), Uv(p), Uv(qq)}), 4.0 * e, acc, 1);
      }
      return acc;
    };
  }
  // normalisation (:177-187), the Wiener weight (:44-46) and the estimate, on the device; AL comes back once, if asked for
  const cx<T>* tot = nullptr;
  const double* dAL_in = q.dev_plane(AL_in, pl);
  if (!AL_in) {
    cx<T>* t = q.new_fourier(1);
    bool first = true;
    for (int i : idx) for (int j : idx) {
      add_to(t, 1.0, lmul(A(i, j), (i == 1) + (j == 1), (i == 2) + (j == 2), true, 1), 1.0, first, 1);
      first = false;
    }
    tot = t;
  }
  const double* dCphi = q.dev_plane(Cphi, pl);
  double* dAL = (double*)q.take(sizeof(double) * pl);
  cx<T>* wd = q.new_fourier(1);
  CMBL_LAUNCH(c, K_HARM, (k_qe_norm<T>), dim3(gpl), 0, c->stream, tot, dAL_in, dCphi, wiener ? 1 : 0, dAL, wd, pl);
  for (int b = 0; b < B; ++b) c->map_fma((T*)(phiqe_ref + (long)b * pl), (const T*)wd, (const T*)(un + (long)b * pl), 1.0, false, 2 * pl);
  if (AL_out) CMBL_HIP(hipMemcpyAsync(AL_out, dAL, sizeof(double) * pl, hipMemcpyDefault, c->stream));     // host or device destination
  CMBL_HIP(hipStreamSynchronize(c->stream));
}

}  // namespace cmbl
