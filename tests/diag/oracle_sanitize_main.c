/* Driver of the sanitizer build of the CPU oracle (tests/test_sanitizers.py): the oracle is the sole judge of every parity test, so its
 * 2,000 lines of index arithmetic over a packed blob are run once under AddressSanitizer + UndefinedBehaviorSanitizer.
 * usage: oracle_sanitize <blob file> <states file (float32 records)> <n steps> [cloth file]    -- test infrastructure only */
#include "../../oracle/agx_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
  void* p = malloc(*n); if (fread(p, 1, *n, f) != *n) exit(2); fclose(f); return p;
}
int main(int argc, char** argv) {
  if (argc < 4) return 2;
  size_t nb, ns, nc = 0;
  uint32_t* blob = (uint32_t*)slurp(argv[1], &nb);
  float* states = (float*)slurp(argv[2], &ns);
  float* cloth = argc > 4 ? (float*)slurp(argv[4], &nc) : NULL;
  const int steps = atoi(argv[3]);
  agxo_model* m = agxo_load(blob, nb / 4);
  if (!m) { fprintf(stderr, "blob rejected\n"); return 3; }
  const int sw = agxo_state_words(m), n = (int)(ns / 4 / sw), nn = agxo_cloth_nodes(m);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    float* s = states + (size_t)i * sw; float* c = cloth ? cloth + (size_t)i * 6 * nn : NULL;
    float obs[256], rew, info[8], act[64]; int done;
    for (int k = 0; k < steps; k++) {
      for (int a = 0; a < 64; a++) act[a] = (float)(((k * 31 + a * 17 + i * 7) % 41) - 20) / 20.0f;
      if (c) agxo_step_cloth(m, s, c, act, obs, &rew, &done, info); else agxo_step(m, s, act, obs, &rew, &done, info);
      sum += rew + obs[0];
    }
    if (!c) { agxo_settle(m, s, 2); double out[96 * 13]; agxo_substep_debug(m, s, out, 96); }
    /* the world API of the reference bridge */
    agxo_world* w = agxo_world_create(m, s, c);
    agxo_world_step(w);
    double cont[16 * 64], p[3], q[4], lin[3], ang[3]; agxo_world_contacts(w, cont, 64); agxo_world_frame(w, 3, 0, p, q, lin, ang);
    agxo_world_store(w, s, c); agxo_world_free(w);
  }
  printf("ok %d records x %d steps, checksum %.6f\n", n, steps, sum);
  agxo_free(m); free(blob); free(states); free(cloth);
  return 0;
}
