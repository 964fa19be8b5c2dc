/* Search for a small comparator network that sorts every 8x8 array of keys x_i + y_j with x and y ascending
 * (keys sorted along rows and columns) -- the input class of the random-overlap mixing step in its normal case
 * (reference: k_rorr, src/radtran/clima_radtran_types.f90:826-852; clima_amd/csrc/kernels.hip rorr_mix8).
 *
 * Zero-one principle for the class: a network sorts every such array iff it sorts the 12 870 monotone 0/1
 * matrices (every threshold image of a row- and column-sorted array is one of them, and each of them is one).
 *
 * State = the set of 0/1 vectors still possible; a compare-exchange (a, b) maps it; it is done when every
 * vector is sorted (zeros on the low wires), optionally modulo the order INSIDE given groups of output wires
 * (the window rebin never looks at the order inside the gaps between its windows).
 * Greedy with random tie-breaking / top-k sampling; the score of a candidate is the number of vectors it
 * merges (|S| shrinks by that), then the total displacement it removes.
 *
 *   netsearch <seed> <restarts> <layout 0|1|2> <groups 0|1> <topk> [out-file]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define N 64
typedef uint64_t u64;

static u64 rng_s;
static inline u64 rnd(void) { rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17; return rng_s; }

static int wire_of[8][8];
static int NW = 64;          /* wires in use */
static int merge_rows = 0;    /* > 0: inputs = (merge_rows x 8) staircases, both halves pre-sorted */   /* matrix element -> wire */
static int group_of[N];     /* output wire -> group id (wires of one group may end in any order) */
static int ngroups;

static int cmp_u64(const void *a, const void *b) { u64 x = *(const u64 *)a, y = *(const u64 *)b; return x < y ? -1 : x > y; }

/* monotone 0/1 matrices: row i has its ones from column c_i on, c_0 >= c_1 >= ... >= c_7 (0..8) */
static int gen_merge_inputs(u64 *out) {
  /* rows r: ones from column c_r on, c_0 >= c_1 >= ...; half sums -> sorted halves (zeros first) */
  int R = merge_rows, half = R * 4, n = 0;
  int c[8];
  for (int r = 0; r < 8; r++) c[r] = 0;
  /* enumerate non-increasing c_0..c_{R-1} in 0..8 */
  int idx[8];
  for (int r = 0; r < R; r++) idx[r] = 0;
  /* recursive enumeration via counters */
  int total = 1;
  for (int r = 0; r < R; r++) total *= 9;
  for (int code = 0; code < total; code++) {
    int t = code, ok = 1;
    for (int r = 0; r < R; r++) { c[r] = t % 9; t /= 9; }
    for (int r = 1; r < R; r++) if (c[r] > c[r - 1]) ok = 0;
    if (!ok) continue;
    int n1 = 0, n2 = 0;
    for (int r = 0; r < R / 2; r++) n1 += 8 - c[r];
    for (int r = R / 2; r < R; r++) n2 += 8 - c[r];
    u64 v = 0;
    for (int w = half - n1; w < half; w++) v |= 1ULL << w;
    for (int w = 2 * half - n2; w < 2 * half; w++) v |= 1ULL << w;
    int dup = 0;
    for (int k = 0; k < n; k++) if (out[k] == v) dup = 1;
    if (!dup) out[n++] = v;
  }
  return n;
}

static int gen_inputs(u64 *out) {
  if (merge_rows) return gen_merge_inputs(out);
  int n = 0, c[8];
  for (c[0] = 0; c[0] <= 8; c[0]++) for (c[1] = 0; c[1] <= c[0]; c[1]++) for (c[2] = 0; c[2] <= c[1]; c[2]++)
  for (c[3] = 0; c[3] <= c[2]; c[3]++) for (c[4] = 0; c[4] <= c[3]; c[4]++) for (c[5] = 0; c[5] <= c[4]; c[5]++)
  for (c[6] = 0; c[6] <= c[5]; c[6]++) for (c[7] = 0; c[7] <= c[6]; c[7]++) {
    u64 v = 0;
    for (int i = 0; i < 8; i++) for (int j = c[i]; j < 8; j++) v |= 1ULL << wire_of[i][j];
    out[n++] = v;
  }
  return n;
}

/* sorted modulo groups: going up the groups, a group of all zeros ... one mixed group ... all ones */
static int is_done_vec(u64 v) {
  int ones = __builtin_popcountll(v), zeros = NW - ones;
  /* wires < zeros must be 0 and the others 1, except inside the group that holds the boundary */
  for (int w = 0; w < NW; w++) {
    int want = w >= zeros;
    int have = (v >> w) & 1;
    if (want != have) {
      /* allowed only if w's group straddles the boundary */
      int g = group_of[w], lo = w, hi = w;
      while (lo > 0 && group_of[lo - 1] == g) lo--;
      while (hi < NW - 1 && group_of[hi + 1] == g) hi++;
      if (!(lo < zeros && zeros <= hi)) return 0;
      /* and then the number of zeros inside the group must be right */
      int zin = 0;
      for (int u = lo; u <= hi; u++) zin += !((v >> u) & 1);
      if (zin != zeros - lo) return 0;
      /* everything outside the group is checked by the loop itself */
    }
  }
  return 1;
}

/* open-addressing hash set */
#define HBITS 16
#define HSIZE (1 << HBITS)
static u64 htab[HSIZE];
static unsigned char hocc[HSIZE];
static inline unsigned hidx(u64 v) { return (unsigned)((v * 0x9E3779B97F4A7C15ULL) >> (64 - HBITS)); }
static void hclear(void) { memset(hocc, 0, sizeof hocc); }
static void hins(u64 v) { unsigned i = hidx(v); while (hocc[i]) { if (htab[i] == v) return; i = (i + 1) & (HSIZE - 1); } hocc[i] = 1; htab[i] = v; }
static inline int hhas(u64 v) { unsigned i = hidx(v); while (hocc[i]) { if (htab[i] == v) return 1; i = (i + 1) & (HSIZE - 1); } return 0; }

static long displacement(const u64 *S, int n) {
  /* sum over vectors of (number of ones below the boundary) -- 0 iff all sorted */
  long d = 0;
  for (int k = 0; k < n; k++) {
    int zeros = NW - __builtin_popcountll(S[k]);
    u64 low = zeros == 64 ? ~0ULL : ((1ULL << zeros) - 1);
    d += __builtin_popcountll(S[k] & low);
  }
  return d;
}

typedef struct { int a, b; } CE;

static int run(int topk, CE *net, int maxlen, int best_known) {
  static u64 S[13000], T[13000];
  int n = gen_inputs(S);
  qsort(S, n, sizeof(u64), cmp_u64);
  int len = 0;
  for (;;) {
    int undone = 0;
    for (int k = 0; k < n; k++) if (!is_done_vec(S[k])) { undone++; }
    if (!undone) return len;
    if (len >= maxlen || len >= best_known) return -1;
    hclear();
    for (int k = 0; k < n; k++) hins(S[k]);
    /* per-wire-pair counts */
    static int merges[N][N], moved[N][N];
    memset(merges, 0, sizeof merges);
    memset(moved, 0, sizeof moved);
    for (int k = 0; k < n; k++) {
      u64 v = S[k];
      u64 ones = v, zeros = ~v & (NW == 64 ? ~0ULL : ((1ULL << NW) - 1));
      /* pairs (a<b) with bit a = 1 and bit b = 0 swap */
      for (u64 oa = ones; oa; oa &= oa - 1) {
        int a = __builtin_ctzll(oa);
        u64 zb = zeros & ~((2ULL << a) - 1);   /* zero bits above a */
        if (a == 63) zb = 0;
        for (; zb; zb &= zb - 1) {
          int b = __builtin_ctzll(zb);
          moved[a][b]++;
          if (hhas(v ^ (1ULL << a) ^ (1ULL << b))) merges[a][b]++;
        }
      }
    }
    /* candidates ranked by (merges, moved*(b-a))  */
    typedef struct { long score; int a, b; } Cand;
    static Cand cand[N * N];
    int nc = 0;
    for (int a = 0; a < N; a++) for (int b = a + 1; b < N; b++) {
      if (!moved[a][b]) continue;
      long sc = (long)merges[a][b] * 100000L + (long)moved[a][b] * (b - a > 32 ? 32 : b - a) + (long)(rnd() % 64);
      cand[nc].score = sc; cand[nc].a = a; cand[nc].b = b; nc++;
    }
    if (!nc) return -1;
    /* pick among the top-k at random (k = 1: pure greedy) */
    int pick = 0;
    if (topk > 1) {
      /* partial selection of the top-k */
      for (int t = 0; t < topk && t < nc; t++) {
        int bi = t;
        for (int u = t + 1; u < nc; u++) if (cand[u].score > cand[bi].score) bi = u;
        Cand tmp = cand[t]; cand[t] = cand[bi]; cand[bi] = tmp;
      }
      int kk = topk < nc ? topk : nc;
      /* geometric preference for the best */
      pick = 0;
      while (pick + 1 < kk && (rnd() % 100) < 35) pick++;
    } else {
      int bi = 0;
      for (int u = 1; u < nc; u++) if (cand[u].score > cand[bi].score) bi = u;
      pick = bi;
    }
    int a = cand[pick].a, b = cand[pick].b;
    net[len].a = a; net[len].b = b; len++;
    u64 m = (1ULL << a) | (1ULL << b);
    int nn = 0;
    for (int k = 0; k < n; k++) {
      u64 v = S[k];
      if (((v >> a) & 1) && !((v >> b) & 1)) v ^= m;
      T[nn++] = v;
    }
    qsort(T, nn, sizeof(u64), cmp_u64);
    n = 0;
    for (int k = 0; k < nn; k++) if (k == 0 || T[k] != T[k - 1]) S[n++] = T[k];
  }
}

int main(int argc, char **argv) {
  u64 seed = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  int restarts = argc > 2 ? atoi(argv[2]) : 1;
  int layout = argc > 3 ? atoi(argv[3]) : 0;
  int groups = argc > 4 ? atoi(argv[4]) : 0;
  int topk = argc > 5 ? atoi(argv[5]) : 1;
  const char *outf = argc > 6 ? argv[6] : NULL;
  rng_s = seed * 0x9E3779B97F4A7C15ULL + 12345;
  /* layouts: 0 = row-major (wire = 8 i + j); 1 = by the middle of the element's possible rank range;
   * 2 = by anti-diagonal then row */
  if (layout >= 10) { merge_rows = layout - 10; NW = merge_rows * 8; layout = 0; }
  if (layout == 0) {
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) wire_of[i][j] = 8 * i + j;
  } else {
    int key[64], idx[64];
    for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) {
      int lo = (i + 1) * (j + 1) - 1, hi = 64 - (8 - i) * (8 - j);
      key[8 * i + j] = layout == 1 ? (lo + hi) * 64 + 8 * i + j : (i + j) * 64 + i;
      idx[8 * i + j] = 8 * i + j;
    }
    for (int a = 0; a < 64; a++) for (int b = a + 1; b < 64; b++) if (key[idx[b]] < key[idx[a]]) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
    for (int w = 0; w < 64; w++) wire_of[idx[w] / 8][idx[w] % 8] = w;
  }
  /* groups: the gaps between the tight rebin windows [5,8] [12,18] [19,28] [27,36] [35,44] [45,51] [55,58]
   * (kernels.hip rb_lo/rb_hi<true>): wires 0-4, 9-11, 52-54, 59-63 may end in any order */
  for (int w = 0; w < N; w++) group_of[w] = 100 + w;
  if (groups) {
    for (int w = 0; w <= 4; w++) group_of[w] = 0;
    for (int w = 9; w <= 11; w++) group_of[w] = 1;
    for (int w = 52; w <= 54; w++) group_of[w] = 2;
    for (int w = 59; w <= 63; w++) group_of[w] = 3;
  }
  ngroups = 4;
  int best = 1000;
  static CE net[2000], bestnet[2000];
  for (int r = 0; r < restarts; r++) {
    int len = run(r == 0 ? 1 : topk, net, 1000, best);
    if (len > 0 && len < best) {
      best = len;
      memcpy(bestnet, net, sizeof(CE) * len);
      fprintf(stderr, "restart %d: %d exchanges\n", r, len);
      if (outf) {
        FILE *f = fopen(outf, "w");
        fprintf(f, "# layout %d groups %d: %d exchanges; wire_of[i][j] rows:\n", layout, groups, len);
        for (int i = 0; i < 8; i++) { fprintf(f, "#W"); for (int j = 0; j < 8; j++) fprintf(f, " %d", wire_of[i][j]); fprintf(f, "\n"); }
        for (int k = 0; k < len; k++) fprintf(f, "%d %d\n", bestnet[k].a, bestnet[k].b);
        fclose(f);
      }
    }
  }
  printf("best %d\n", best);
  return 0;
}
