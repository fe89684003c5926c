// C++ counterpart of the reference's demo caf_rust/src/main.rs:10-32, with its TODO done:
// the two .c64 files may be given as arguments (main.rs:1-2).
#include <cstdio>

#include "caf_hip.hpp"

int main(int argc, char **argv)
{
    using namespace caf;
    const std::string needle_file = argc > 2 ? argv[1] : "tests/golden/data/chirp_0_raw.c64";
    const std::string haystack_file = argc > 2 ? argv[2] : "tests/golden/data/chirp_0_T+202samp_F+69.25Hz.c64";
    // needle = the clean burst, haystack = the delayed, Doppler-shifted capture, cut or padded to the needle's length
    auto needle = read_file_c64(needle_file);
    auto haystack = read_file_c64(haystack_file);
    haystack.resize(needle.size(), Complex64(0.0, 0.0));
    // the Doppler grid: 400 shifts built from integer milli-hertz so that every value is the exactly rounded double
    std::vector<double> shifts;
    for (int m = -100000; m < 100000; m += 500) shifts.push_back(m / 1e3);
    // all rows on the GPU, then the first strictly greater row peak
    auto surface = CafHip::caf_surface(needle, haystack, shifts, 48000);
    auto peak = CafHip::find_peak(std::move(surface));
    // same two lines as the reference's demo prints (sample index / 48 = milliseconds at 48 kHz)
    std::printf("Frequency offset: %.1fHz\n", peak.first);
    std::printf("Time offset: %zu samples (%.3fms)\n", peak.second, (double)peak.second / 48.0);
    return 0;
}
