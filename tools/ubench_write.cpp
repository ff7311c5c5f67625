// Micro-benchmark (host only): how fast can T threads put N bytes into ONE file on this box, and by which route?
//   g++ -O2 -pthread tools/ubench_write.cpp -o /tmp/ubench_write && /tmp/ubench_write /dev/shm 4 16
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const std::string dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t gb = argc > 2 ? atoi(argv[2]) : 4;
    const int T = argc > 3 ? atoi(argv[3]) : 16;
    const size_t N = gb << 30, per = N / T;
    std::vector<char> src(64 << 20, 'A');
    const std::string path = dir + "/pc_ubench_write.bin";
    auto run = [&](const char *name, auto body) {
        unlink(path.c_str());
        const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
        const double t0 = now();
        body(fd);
        close(fd);
        const double dt = now() - t0;
        printf("%-44s %6.2f s  %6.2f GB/s\n", name, dt, N / 1e9 / dt);
        fflush(stdout);
    };
    auto par = [&](auto fn) { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(fn, t); for (auto &x : th) x.join(); };
    run("pwrite, 8 MB calls, T threads", [&](int fd) {
        par([&](int t) { for (size_t o = 0; o < per; o += 8 << 20) pwrite(fd, src.data(), 8 << 20, t * per + o); });
    });
    run("pwrite, 8 MB calls, 1 thread", [&](int fd) {
        for (size_t o = 0; o < N; o += 8 << 20) pwrite(fd, src.data(), 8 << 20, o);
    });
    run("ftruncate + mmap shared + memcpy, T threads", [&](int fd) {
        ftruncate(fd, N);
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        par([&](int t) { for (size_t o = 0; o < per; o += 8 << 20) memcpy(m + t * per + o, src.data(), 8 << 20); });
        munmap(m, N);
    });
    run("fallocate + mmap shared + memcpy, T threads", [&](int fd) {
        posix_fallocate(fd, 0, N);
        char *m = (char *)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        par([&](int t) { for (size_t o = 0; o < per; o += 8 << 20) memcpy(m + t * per + o, src.data(), 8 << 20); });
        munmap(m, N);
    });
    run("parallel fallocate per thread, then pwrite", [&](int fd) {
        par([&](int t) { fallocate(fd, 0, t * per, per); });
        par([&](int t) { for (size_t o = 0; o < per; o += 8 << 20) pwrite(fd, src.data(), 8 << 20, t * per + o); });
    });
    run("per-thread mmap MAP_POPULATE + memcpy", [&](int fd) {
        ftruncate(fd, N);
        par([&](int t) {
            char *m = (char *)mmap(nullptr, per, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_POPULATE, fd, t * per);
            for (size_t o = 0; o < per; o += 8 << 20) memcpy(m + o, src.data(), 8 << 20);
            munmap(m, per);
        });
    });
    run("T separate files, pwrite", [&](int) {
        par([&](int t) {
            const std::string p2 = path + "." + std::to_string(t);
            const int f2 = open(p2.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0666);
            for (size_t o = 0; o < per; o += 8 << 20) pwrite(f2, src.data(), 8 << 20, o);
            close(f2); unlink(p2.c_str());
        });
    });
    unlink(path.c_str());
    return 0;
}
