// Host-only check of rato_common.h's read-back helpers (no GPU call is made on the paths exercised here): a buffer is armed,
// a second thread plays the kernel and writes the words one by one with pauses, the waiter must return only when every
// word has arrived -- including words whose value is a NaN, zero, or the sentinel's neighbour.
#include <stdio.h>
#include <string.h>

#include <thread>
#include <vector>

#include "rato_common.h"

int main() {
  const int n = 115;
  std::vector<double> buf(n, 1.0);
  for (int round = 0; round < 20; ++round) {
    rato::readback_arm(buf.data(), n);
    for (int i = 0; i < n; ++i) {
      uint64_t w;
      memcpy(&w, &buf[i], 8);
      if (w != rato::READBACK_PENDING) return 2;
    }
    std::vector<double> vals(n);
    for (int i = 0; i < n; ++i) vals[i] = (i % 7 == 0) ? __builtin_nan("") : (i % 5 == 0 ? 0.0 : 0.5 * i - round);
    uint64_t near = rato::READBACK_PENDING ^ 1ull;   // a different NaN payload is a legitimate value
    memcpy(&vals[3], &near, 8);
    std::thread producer([&] {
      for (int i = n - 1; i >= 0; --i) {              // (reverse order: the first words arrive last)
        if (i % 16 == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
        volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(buf.data());
        uint64_t v;
        memcpy(&v, &vals[i], 8);
        w[i] = v;
      }
    });
    const hipError_t e = rato::readback_wait(buf.data(), n, nullptr);
    int missing = 0;
    for (int i = 0; i < n; ++i) missing += memcmp(&buf[i], &vals[i], 8) != 0;
    producer.join();
    if (e != hipSuccess) return 3;
    if (missing) return 4;                            // returned before every word was there
  }
  printf("readback ok\n");
  return 0;
}
