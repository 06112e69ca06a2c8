// Known-answer generator for the negative-sample index stream.
// The reference's UniformGenerator (cymf/math.pyx:12-18) IS libstdc++'s
//   std::mt19937(seed) + std::uniform_int_distribution<long>(a, b-1)
// and is a cdef class (not callable from Python), so the fixtures are produced by calling
// that third-party dependency directly: GCC 11.4 libstdc++, the version the oracle/_ref build
// of the reference links.  Our own code, not copied from the reference.
//   usage: gen_stream <seed> <range> <count>   -> one draw per line on stdout
#include <cstdio>
#include <cstdlib>
#include <random>
int main(int argc, char** argv) {
    if (argc < 4) return 1;
    unsigned seed = (unsigned)std::strtoul(argv[1], nullptr, 10);
    long range = std::strtol(argv[2], nullptr, 10);
    long n = std::strtol(argv[3], nullptr, 10);
    std::mt19937 rng(seed);
    std::uniform_int_distribution<long> uni(0, range - 1);
    for (long i = 0; i < n; ++i) std::printf("%ld\n", uni(rng));
    return 0;
}
