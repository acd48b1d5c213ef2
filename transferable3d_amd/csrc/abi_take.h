// Growable argument structs of the C ABI (include/t3d.h: `struct_size`).  No HIP dependency: tests/test_abi.py compiles this header
// with g++ against a struct in an "older" and a "newer" declaration.
#pragma once
#include <stdint.h>
#include <string.h>

// `a` points at a caller's struct whose first field is `uint32_t struct_size` = the caller's sizeof.  The library's declaration is T.
//   struct_size == sizeof(T)              the caller was built against this header: used in place
//   v2_size <= struct_size < sizeof(T)    an OLDER caller (fields were appended since): its bytes are copied into `local`, every field
//                                         it does not know reads 0 -- the documented default of every appended field -- and `a` is
//                                         redirected to the copy
//   struct_size > sizeof(T)               a NEWER caller: accepted when every byte beyond sizeof(T) is 0 (it asks for nothing this
//                                         library does not know), refused otherwise
//   struct_size < v2_size                 not a struct of ABI version 2 or later: refused
// Returns 0 or -4 (T3D_ERR_ABI).  A null `a` is left to the entry point's own argument check.
template <class T>
static inline int t3d_abi_take(const T*& a, T& local, uint32_t v2_size) {
  if (a == nullptr) return 0;
  const uint32_t n = a->struct_size;
  if (n == sizeof(T)) return 0;
  if (n < v2_size) return -4;
  if (n > sizeof(T)) {
    const unsigned char* tail = reinterpret_cast<const unsigned char*>(a) + sizeof(T);
    for (uint32_t i = 0; i < n - (uint32_t)sizeof(T); ++i)
      if (tail[i] != 0) return -4;
    memcpy(&local, a, sizeof(T));
  } else {
    memset(&local, 0, sizeof(T));
    memcpy(&local, a, n);
  }
  local.struct_size = (uint32_t)sizeof(T);
  a = &local;
  return 0;
}
// at the head of an entry point:  T3D_ABI_TAKE(pointmlp_fwd_args, a);   (t3d_<name>, T3D_V2_SIZE_<name> of t3d.h)
#define T3D_ABI_TAKE(name, a)                                                              \
  t3d_##name a##_abi_local_;                                                               \
  do {                                                                                     \
    const int abi_e_ = t3d_abi_take(a, a##_abi_local_, (uint32_t)T3D_V2_SIZE_##name);      \
    if (abi_e_ != 0) return abi_e_;                                                        \
  } while (0)
