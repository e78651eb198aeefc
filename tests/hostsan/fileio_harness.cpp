// Host-only part of the product (csrc/fileio.cpp: WorldMi355WriteFiles, no HIP) under AddressSanitizer +
// UndefinedBehaviorSanitizer: tests/test_sanitizers.py compiles this file together with fileio.cpp by g++ and runs it.
// Writes N files of different sizes from one buffer with T threads, reads them back, then the error path (a directory
// that does not exist must come back as WM_ERR_IO with the file's name in the message).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "world_mi355.h"

namespace wm {
static std::string g_msg;
void set_error(const char* msg) { g_msg = msg; }
}  // namespace wm

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  const int n = 257, threads = 7;
  std::vector<float> slab(200000);
  for (size_t i = 0; i < slab.size(); ++i) slab[i] = (float)i * 0.5f;
  std::vector<std::string> names;
  std::vector<const char*> paths;
  std::vector<const void*> data;
  std::vector<size_t> bytes;
  size_t at = 0;
  for (int i = 0; i < n; ++i) {
    names.push_back(dir + "/f" + std::to_string(i) + ".bin");
    const size_t cnt = (size_t)(i * 37 % 1500);                  // some files are empty
    data.push_back(slab.data() + at);
    bytes.push_back(cnt * sizeof(float));
    at += cnt;
  }
  for (auto& s : names) paths.push_back(s.c_str());
  if (WorldMi355WriteFiles(n, paths.data(), data.data(), bytes.data(), threads) != WM_OK) return 3;
  for (int i = 0; i < n; ++i) {
    FILE* f = fopen(paths[i], "rb");
    if (!f) return 4;
    std::vector<char> got(bytes[i] + 1);
    const size_t r = fread(got.data(), 1, bytes[i] + 1, f);
    fclose(f);
    if (r != bytes[i] || memcmp(got.data(), data[i], bytes[i]) != 0) return 5;
  }
  // more threads than files, zero files, bad arguments
  if (WorldMi355WriteFiles(2, paths.data(), data.data(), bytes.data(), 64) != WM_OK) return 6;
  if (WorldMi355WriteFiles(0, nullptr, nullptr, nullptr, 4) != WM_OK) return 7;
  if (WorldMi355WriteFiles(3, nullptr, data.data(), bytes.data(), 4) != WM_ERR_BAD_ARG) return 8;
  const std::string bad = dir + "/no/such/dir/x.bin";
  paths[5] = bad.c_str();
  if (WorldMi355WriteFiles(n, paths.data(), data.data(), bytes.data(), threads) != WM_ERR_IO) return 9;
  if (wm::g_msg.find("no/such/dir/x.bin") == std::string::npos) return 10;
  printf("fileio ok\n");
  return 0;
}
