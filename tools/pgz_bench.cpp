// g++ -O2 -std=c++17 -I drprg_amd/csrc -o /tmp/pgz_bench tools/pgz_bench.cpp drprg_amd/csrc/pgunzip.cpp -lz -ldl -lpthread
// /tmp/pgz_bench file.gz THREADS [CHUNK_BYTES]: wall time of the parallel gunzip alone (text discarded)
#include "pgunzip.h"
#include <chrono>
#include <cstdio>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <vector>
using namespace drprg;
int main(int argc, char** argv) {
    int fd = open(argv[1], O_RDONLY); struct stat sb; fstat(fd, &sb);
    void* m = mmap(nullptr, sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    int th = atoi(argv[2]);
    std::vector<char> buf(size_t(32) << 20);
    for (int rep = 0; rep < 2; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    ParallelGunzip pg((const unsigned char*)m, sb.st_size, th, argc > 3 ? atol(argv[3]) : 0);
    size_t total = 0;
    for (size_t n; (n = pg.read(buf.data(), buf.size())) > 0;) total += n;
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("threads %d: %zu bytes %.3f s %.0f MB/s accepted %lu redone %lu\n", th, total, dt, total / dt / 1e6, pg.chunks_accepted(), pg.chunks_redone());
    }
}
