// C wrappers around sdrx_amd/csrc/fused_index.hpp for tests/test_fused_index.py (CPU): built with g++, loaded with ctypes.
#include "../../sdrx_amd/csrc/fused_index.hpp"
using namespace sdrxgpu::fx;
extern "C" {
int fx_locate(int k25, int r0m6, int s, int Rrel, int *n_c) { return locate(k25, r0m6, s, Rrel, n_c); }
int fx_overlaps(int k25, int n_y1, int s, int Rrel) { return overlaps(k25, n_y1, s, Rrel) ? 1 : 0; }
int fx_entry_output(int k25, int r0m6, int n_y1, int s, int entry, int *Rrel, int *n_c) { return entry_output(k25, r0m6, n_y1, s, entry, Rrel, n_c) ? 1 : 0; }
int fx_emits(int kph, int n, int *slot) { return emits(kph, n, slot) ? 1 : 0; }
int fx_sigma(int c) { return sigma_of(c); }
int fx_delta(int c) { return delta_of(c); }
}
