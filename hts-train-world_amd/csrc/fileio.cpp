// fileio.cpp -- the feature files of a batch written by native threads (host only, no HIP).
//
// The reference's `analysis` CLI ends with three fwrite loops per utterance (test/analysis.cpp:360-390); the recipe
// runs it once per utterance.  A batched sweep hands back thousands of small arrays at once (configs[3]: 3 000 files
// of about 100 KB per pass when coded): from a Python thread pool every file cost about a millisecond of
// interpreter time under the global lock, more than the analysis of the whole corpus.  WorldMi355WriteFiles takes
// the whole list in one call (ctypes drops the lock for its duration) and works it off with plain threads.
#include <errno.h>
#include <fcntl.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "../../include/world_mi355.h"

namespace wm {
void set_error(const char* msg);
}

extern "C" int WorldMi355WriteFiles(int n_files, const char* const* paths, const void* const* data,
                                    const size_t* bytes, int n_threads) {
  if (n_files < 0 || (n_files > 0 && (!paths || !data || !bytes))) {
    wm::set_error("WriteFiles: bad argument");
    return WM_ERR_BAD_ARG;
  }
  if (n_files == 0) return WM_OK;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_files) n_threads = n_files;
  std::atomic<int> next(0), failed(-1);
  std::atomic<int> err(0);
  auto work = [&]() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n_files) return;
      const int fd = open(paths[i], O_WRONLY | O_CREAT | O_TRUNC, 0644);
      int e = 0;
      if (fd < 0) {
        e = errno;
      } else {
        const char* p = static_cast<const char*>(data[i]);
        size_t left = bytes[i];
        while (left > 0) {
          const ssize_t w = write(fd, p, left);
          if (w < 0) {
            if (errno == EINTR) continue;
            e = errno;
            break;
          }
          p += w;
          left -= (size_t)w;
        }
        if (close(fd) != 0 && !e) e = errno;
      }
      if (e) {
        int none = -1;
        if (failed.compare_exchange_strong(none, i)) err.store(e);
      }
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (failed.load() >= 0) {
    const std::string msg = std::string("WriteFiles: ") + paths[failed.load()] + ": " + strerror(err.load());
    wm::set_error(msg.c_str());
    return WM_ERR_IO;
  }
  return WM_OK;
}
