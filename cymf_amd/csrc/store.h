// store.h -- float64 host <-> typed device factor storage (defined in bpr.hip).
#pragma once
#include "common.h"
#include "rng.h"
#include "rows.h"

namespace cymf {
template <typename T> int upload_f64(DevBuf<T> &dst, const double *src, size_t n, hipStream_t s);
template <typename T> int download_f64(const DevBuf<T> &src, double *dst, size_t n, hipStream_t s);
template <typename T> int fill_dev(DevBuf<T> &b, size_t n, T v, hipStream_t s);
}  // namespace cymf
