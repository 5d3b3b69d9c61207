// tools/asan_tfd_host.cpp -- the TFD ladder's host code (csrc/fc_tfd_host.cpp: the all-host ladder, and the device ladder's
// routines of fc_tfd_core.h run by one host thread) under AddressSanitizer / UBSan or ThreadSanitizer on the CPU.
//   hipcc -x hip --cuda-host-only -O1 -g -fsanitize=address,undefined -std=c++17 tools/asan_tfd_host.cpp -o /tmp/asan_tfd -lpthread
//   (or -fsanitize=thread); /tmp/asan_tfd [n] [kind: stars|chain|mixed|near] [seed]
// The shapes are tests/test_tfd_gpu_graph.py's (_random_first_match); 700000 stars 3 is the case of round 3's abort.
#include <random>
#include "../firecode_amd/csrc/fc_tfd_host.cpp"

namespace fc {  // what fc_tfd_host.cpp takes from the other translation units
std::string &last_error() { static thread_local std::string s; return s; }
int set_error(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}
Context &ctx() { static Context c; return c; }
void *pool_take(size_t, size_t *, bool) { return nullptr; }
void pool_give(void *, size_t) {}
bool host_memory_is_pinned(const void *) { return false; }
bool staged_uploads() { return false; }
int h2d_staged(void *, const void *, size_t, hipStream_t) { return FC_E_HIP; }
int d2h_staged(void *, const void *, size_t, hipStream_t) { return FC_E_HIP; }
int tfd_ladder_device(const int64_t *, const int64_t *, int64_t, uint8_t *) { return FC_E_HIP; }
}  // namespace fc

int main(int argc, char **argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 700000;
  const std::string kind = argc > 2 ? argv[2] : "stars";
  const unsigned seed = argc > 3 ? (unsigned)atoi(argv[3]) : 3;
  std::mt19937_64 rng(seed);
  std::vector<int64_t> fm((size_t)n, -1);
  std::vector<int64_t> hubs;
  if (kind == "stars") {
    std::vector<char> is((size_t)n, 0);
    for (int64_t k = 0; k < std::max<int64_t>(2, n / 300); ++k) is[(size_t)(rng() % (uint64_t)n)] = 1;
    for (int64_t i = 0; i < n; ++i)
      if (is[(size_t)i]) hubs.push_back(i);
  }
  for (int64_t i = 0; i + 1 < n; ++i) {
    int64_t j;
    if (kind == "near") j = i + 1 + (int64_t)(rng() % 39);
    else if (kind == "mixed") j = i + ((rng() % 10) < 6 ? 1 + (int64_t)(rng() % 49) : 1 + (int64_t)(rng() % (uint64_t)(n - 1)));
    else if (kind == "stars") {
      auto it = std::lower_bound(hubs.begin(), hubs.end(), i + 1);
      j = it == hubs.end() ? hubs.back() : *it;
    } else j = i + 1;
    if (j > i && j < n && (rng() % 100) < 97) fm[(size_t)i] = j;
  }
  std::vector<uint8_t> a((size_t)n), b((size_t)n);
  for (const char *threads : {"1", "5", "16"}) {
    setenv("FC_TFD_THREADS", threads, 1);
    if (fc::tfd_ladder_host_only(fm.data(), n, a.data()) != FC_OK) {
      fprintf(stderr, "host ladder failed: %s\n", fc::last_error().c_str());
      return 1;
    }
  }
  if (fc::tfd_ladder_emulate_device(fm.data(), n, b.data()) != FC_OK) {
    fprintf(stderr, "emulated device ladder failed: %s\n", fc::last_error().c_str());
    return 1;
  }
  int64_t kept = 0, diff = 0;
  for (int64_t i = 0; i < n; ++i) kept += a[(size_t)i], diff += a[(size_t)i] != b[(size_t)i];
  printf("n %lld kind %s seed %u: kept %lld, host ladder vs emulated device ladder differ in %lld\n", (long long)n, kind.c_str(), seed,
         (long long)kept, (long long)diff);
  return diff != 0;
}
