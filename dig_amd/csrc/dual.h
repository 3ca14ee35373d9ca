// Forward-mode dual numbers for the DERIVATIVE kernels of the pointwise geometry / basis functions
// (csrc/diffgeom.hip).  energy_and_force training (method/run.py:126-131: force = -dE/dpos with create_graph, then
// loss.backward()) differentiates the xyz -> (dist, angle, torsion) -> basis pipeline TWICE; instead of hand-derived
// Hessians of atan2(|a x b|, a.b) and of the dihedral, every pointwise function is written once as a template over
// its scalar type and instantiated with
//     D1<double>        value + one directional derivative            -> gradients (one pass per input direction)
//     D1<D1<double>>    + the derivative of that derivative along a   -> Hessian-vector products g^T H w and J w
//                         second direction
// All derivative arithmetic is float64: these kernels do not have to reproduce the reference's float32 rounding (the
// forward VALUES still come from the bit-exact float32 kernels in geometry.hip / basis.hip), they have to be accurate.
#pragma once
#include <hip/hip_runtime.h>

template <class B>
struct D1 {
  B v, d;
};

// ---- scalar leaves ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double dl_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double dl_sin(double x) { return sin(x); }
__device__ __forceinline__ double dl_cos(double x) { return cos(x); }
__device__ __forceinline__ double dl_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ double dl_val(double x) { return x; }
template <class B>
__device__ __forceinline__ double dl_val(const D1<B>& x) { return dl_val(x.v); }

// lift a plain number into the dual type (all derivative parts zero)
template <class T>
struct Lift;
template <>
struct Lift<double> {
  __device__ __forceinline__ static double of(double x) { return x; }
};
template <class B>
struct Lift<D1<B>> {
  __device__ __forceinline__ static D1<B> of(double x) { return {Lift<B>::of(x), Lift<B>::of(0.0)}; }
};

// ---- arithmetic ------------------------------------------------------------------------------------------------
template <class B>
__device__ __forceinline__ D1<B> operator+(const D1<B>& a, const D1<B>& b) { return {a.v + b.v, a.d + b.d}; }
template <class B>
__device__ __forceinline__ D1<B> operator-(const D1<B>& a, const D1<B>& b) { return {a.v - b.v, a.d - b.d}; }
template <class B>
__device__ __forceinline__ D1<B> operator-(const D1<B>& a) { return {-a.v, -a.d}; }
template <class B>
__device__ __forceinline__ D1<B> operator*(const D1<B>& a, const D1<B>& b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
template <class B>
__device__ __forceinline__ D1<B> operator/(const D1<B>& a, const D1<B>& b) {
  B q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
template <class B>
__device__ __forceinline__ D1<B> operator*(double s, const D1<B>& a) { return {s * a.v, s * a.d}; }
template <class B>
__device__ __forceinline__ D1<B> operator*(const D1<B>& a, double s) { return {a.v * s, a.d * s}; }
template <class B>
__device__ __forceinline__ D1<B> operator/(const D1<B>& a, double s) { return {a.v / s, a.d / s}; }
template <class B>
__device__ __forceinline__ D1<B> operator/(double s, const D1<B>& a) {
  B q = s / a.v;
  return {q, -(q * a.d) / a.v};
}
template <class B>
__device__ __forceinline__ D1<B> operator+(const D1<B>& a, double s) { return {a.v + s, a.d}; }
template <class B>
__device__ __forceinline__ D1<B> operator+(double s, const D1<B>& a) { return {a.v + s, a.d}; }
template <class B>
__device__ __forceinline__ D1<B> operator-(const D1<B>& a, double s) { return {a.v - s, a.d}; }
template <class B>
__device__ __forceinline__ D1<B> operator-(double s, const D1<B>& a) { return {s - a.v, -a.d}; }

// ---- elementary functions ----------------------------------------------------------------------------------------
template <class B>
__device__ __forceinline__ D1<B> dl_sqrt(const D1<B>& x) {
  B s = dl_sqrt(x.v);
  return {s, x.d / (2.0 * s)};
}
template <class B>
__device__ __forceinline__ D1<B> dl_sin(const D1<B>& x) { return {dl_sin(x.v), dl_cos(x.v) * x.d}; }
template <class B>
__device__ __forceinline__ D1<B> dl_cos(const D1<B>& x) { return {dl_cos(x.v), -(dl_sin(x.v) * x.d)}; }
template <class B>
__device__ __forceinline__ D1<B> dl_atan2(const D1<B>& y, const D1<B>& x) {
  return {dl_atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v)};
}

// 3-vectors over any scalar type
template <class T>
struct V3 {
  T x, y, z;
};
template <class T>
__device__ __forceinline__ V3<T> v3_cross(const V3<T>& a, const V3<T>& b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class T>
__device__ __forceinline__ T v3_dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T>
__device__ __forceinline__ T v3_len(const V3<T>& a) { return dl_sqrt(v3_dot(a, a)); }
