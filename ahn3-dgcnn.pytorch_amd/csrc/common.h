// common.h -- shared device/host helpers for libdgcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>
#include <utility>

#include "dgcnn_hip.h"

namespace dgcnn {

// thread-local error text behind dgcnn_last_error()
void set_error(const char *fmt, ...);

#define DG_FAIL(...)                   \
    do {                               \
        ::dgcnn::set_error(__VA_ARGS__); \
        return 1;                      \
    } while (0)

#define DG_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) DG_FAIL(__VA_ARGS__); \
    } while (0)

// launch check: launches are asynchronous, so this only catches configuration
// errors (bad grid, too much LDS); kernel faults surface at the caller's sync.
#define DG_CHECK_LAUNCH(name)                                                 \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) DG_FAIL("%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// row of accumulator register r (0..15) in a 32x32 MFMA C/D tile for lane-half h:
//   row = (r & 3) + 8 * (r >> 2) + 4 * h      (col = lane & 31)
__device__ __forceinline__ constexpr int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), fully unrolled with constant indices
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace dgcnn
