/* Host-side probe of csrc/tile_layout.h: exposes tile_src() so that tests/test_tile_layout.py can check that the
 * re-layout is a byte permutation (a bijection tile <-> 16 rows x 1 unit) and that every 16-byte lane fragment the
 * matrix-core kernel loads is where the header says it is. */
#include "tile_layout.h"
int probe_unit_bytes(int type) { return mi_unit_bytes(type); }
int probe_tile_bytes(int type) { return mi_tile_bytes(type); }
void probe_tile_src(int type, int b, int * n, int * sb) { int nn, s; tile_src(type, b, nn, s); *n = nn; *sb = s; }
