// write_bench.cpp -- what the box's file system takes: BYTES written to one new file from page-locked-like memory by T
// threads, as gpumap_mapper's writer does (pwrite of disjoint slices), and variants (fallocate first, mmap + memcpy,
// one file per thread).  Host code only.  tools/_build/write_bench DIR GIB THREADS
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
   const std::string dir = argc > 1 ? argv[1] : "/tmp";
   const size_t bytes = (size_t)(atof(argc > 2 ? argv[2] : "4") * (1u << 30));
   const int T = argc > 3 ? atoi(argv[3]) : 16;
   const size_t chunk = (size_t)400 << 20;                    // the writer gets the lines chunk by chunk
   std::vector<char> buf(chunk, 'x');
   for (size_t i = 29; i < chunk; i += 30) buf[i] = '\n';
   for (const char *mode : {"pwrite", "fallocate+pwrite", "mmap", "files"}) {
      const std::string path = dir + "/write_bench.tmp";
      int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
      if (fd < 0) { perror("open"); return 1; }
      const double t0 = now_s();
      if (!strcmp(mode, "mmap") && ftruncate(fd, (off_t)bytes) != 0) { perror("ftruncate"); return 1; }
      for (size_t off = 0; off < bytes; off += chunk) {
         const size_t n = std::min(chunk, bytes - off);
         if (!strcmp(mode, "fallocate+pwrite") && posix_fallocate(fd, (off_t)off, (off_t)n) != 0) { perror("fallocate"); break; }
         char *map = nullptr;
         if (!strcmp(mode, "mmap")) { map = (char *)mmap(nullptr, n, PROT_WRITE, MAP_SHARED, fd, (off_t)off); if (map == MAP_FAILED) { perror("mmap"); return 1; } }
         std::vector<std::thread> th;
         for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
               const size_t a = n * (size_t)t / (size_t)T, e = n * (size_t)(t + 1) / (size_t)T;
               if (map) { memcpy(map + a, buf.data() + a, e - a); return; }
               int f = fd;
               off_t at = (off_t)(off + a);
               if (!strcmp(mode, "files")) { f = open((path + "." + std::to_string(t)).c_str(), O_WRONLY | O_CREAT | (off == 0 ? O_TRUNC : O_APPEND), 0666); at = -1; }
               size_t done = a;
               while (done < e) { ssize_t g = at >= 0 ? pwrite(f, buf.data() + done, e - done, at) : write(f, buf.data() + done, e - done); if (g <= 0) break; done += (size_t)g; if (at >= 0) at += g; }
               if (f != fd) close(f);
            });
         for (auto &x : th) x.join();
         if (map) munmap(map, n);
      }
      const double dt = now_s() - t0;
      printf("{\"mode\": \"%s\", \"dir\": \"%s\", \"GiB\": %.1f, \"threads\": %d, \"seconds\": %.3f, \"GB_per_s\": %.2f}\n", mode, dir.c_str(), bytes / 1073741824.0, T, dt, bytes / dt / 1e9);
      fflush(stdout);
      close(fd);
      unlink(path.c_str());
      for (int t = 0; t < T; t++) unlink((path + "." + std::to_string(t)).c_str());
   }
   return 0;
}
