// Does rocprim::radix_sort_pairs (u32 keys, i64 values) keep the value set for a bit window [begin, end)?
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
int main() {
  const int begins[] = {0, 8, 12, 15, 16};
  for (int win = 0; win < 5; ++win) {
    const int begin = begins[win], end = begin + 16;
    const size_t n = 100000;
    std::vector<uint32_t> k(n);
    std::vector<int64_t> v(n);
    uint32_t s = 12345;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; k[i] = s; v[i] = (int64_t)i + 7; }
    uint32_t *ki, *ko; int64_t *vi, *vo; void* tmp; size_t tb = 0;
    hipMalloc(&ki, n * 4); hipMalloc(&ko, n * 4); hipMalloc(&vi, n * 8); hipMalloc(&vo, n * 8);
    hipMemcpy(ki, k.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(vi, v.data(), n * 8, hipMemcpyHostToDevice);
    hipMemset(vo, 0, n * 8);
    rocprim::radix_sort_pairs(nullptr, tb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int64_t*)nullptr, (int64_t*)nullptr, n, begin, end, (hipStream_t)0);
    size_t tb2 = 0;
    rocprim::radix_sort_pairs(nullptr, tb2, (const uint32_t*)ki, ko, (const int64_t*)vi, vo, n, begin, end, (hipStream_t)0);
    hipMalloc(&tmp, std::max(tb, tb2));
    size_t t = tb;
    hipError_t e = rocprim::radix_sort_pairs(tmp, t, (const uint32_t*)ki, ko, (const int64_t*)vi, vo, n, begin, end, (hipStream_t)0);
    hipDeviceSynchronize();
    std::vector<int64_t> r(n);
    hipMemcpy(r.data(), vo, n * 8, hipMemcpyDeviceToHost);
    std::sort(r.begin(), r.end());
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) bad += r[i] != (int64_t)i + 7;
    printf("window [%d, %d): temp query (null ptrs) %zu, (real ptrs) %zu, rc %d, values lost %zu\n", begin, end, tb, tb2, (int)e, bad);
  }
  return 0;
}
