"""`nbm_fastdiv` (birdsoundclassif_amd/csrc/nbm_fastdiv.h): the host-prepared multiplier / shifts that replace every run-time integer
division of the kernels' row decodes must give n / d EXACTLY -- a wrong quotient is a wrong pixel, silently.  The header's host part is
compiled here with g++ and checked against the machine's division: every divisor the model's geometries produce (and their neighbours),
powers of two and their neighbours, the extremes; for each a strided sweep of the whole 32-bit range, the multiples of d with their
neighbours, and random operands."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = textwrap.dedent(r'''
    #include <cstdio>
    #include <cstdint>
    #include <vector>
    #include "nbm_fastdiv.h"
    int main() {
      std::vector<unsigned> ds;
      for (unsigned d = 1; d <= 4100; ++d) ds.push_back(d);
      const unsigned dims[] = {375, 1024, 188, 512, 94, 256, 47, 128, 24, 64, 12, 32, 6, 16};
      for (int i = 0; i + 1 < 14; i += 2) {
        const unsigned h = dims[i], w = dims[i + 1];
        for (unsigned b = 1; b <= 128; b *= 2) { ds.push_back(h * w); ds.push_back(h * w * b); ds.push_back(((h + 1) / 2) * ((w + 1) / 2)); }
      }
      for (int l = 1; l < 32; ++l) { ds.push_back(1u << l); ds.push_back((1u << l) - 1); ds.push_back((1u << l) + 1); }
      ds.push_back(0x7fffffffu); ds.push_back(0x80000000u); ds.push_back(0x80000001u); ds.push_back(0xfffffffeu); ds.push_back(0xffffffffu);
      unsigned long long bad = 0, checked = 0;
      uint64_t rng = 0x9e3779b97f4a7c15ull;
      for (unsigned d : ds) {
        const nbm_fastdiv f = nbm_fastdiv_make(d);
        auto chk = [&](unsigned n) { ++checked; if (nbm_fdiv_host(n, f) != n / d) { if (bad < 5) printf("n=%u d=%u got %u\n", n, d, nbm_fdiv_host(n, f)); ++bad; } };
        for (uint64_t n = 0; n < (1ull << 32); n += 1000003ull) chk((unsigned)n);
        for (uint64_t k = 0; k < 2000; ++k) {
          const uint64_t m = k * (uint64_t)d;
          if (m > 0xffffffffull) break;
          chk((unsigned)m); if (m) chk((unsigned)(m - 1)); if (m < 0xffffffffull) chk((unsigned)(m + 1));
        }
        const uint64_t top = (0xffffffffull / d) * d;
        chk((unsigned)top); if (top) chk((unsigned)(top - 1)); chk(0xffffffffu); chk(0x80000000u); chk(0x7fffffffu);
        for (int i = 0; i < 4000; ++i) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; chk((unsigned)(rng >> 16)); }
      }
      printf("divisors %zu checked %llu bad %llu\n", ds.size(), checked, bad);
      return bad != 0;
    }
''')


def test_fastdiv_is_exact(tmp_path):
    src = tmp_path / 'fastdiv_check.cpp'
    src.write_text(SRC)
    exe = str(tmp_path / 'fastdiv_check')
    subprocess.run(['g++', '-O2', '-std=c++17', '-I' + os.path.join(ROOT, 'birdsoundclassif_amd', 'csrc'), str(src), '-o', exe],
                   check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and ' bad 0' in out.stdout, out.stdout + out.stderr
