/*
 * oracle.c -- CPU restatement of the per-read match/count path of
 * Roco-scientist/NGS-Barcode-Count (crate barcode-count v0.11.1).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Deliberately string-level and
 * naive: it follows the reference statement by statement, including its
 * quirks, so that the HIP engine can be checked against it bit-exactly.
 * No performance work is done here on purpose.
 *
 * PARITY PINNING: fix_error and MaxSeqErrors are pinned by the reference's
 * doctest values (src/parse.rs:540-551, src/info.rs:479-611).  Everything
 * else on the path has no reference test: parity unpinned by the reference,
 * cross-checked by tests/pyref.py (independent restatement) and
 * tests/golden/ (hand-derived vectors).
 *
 * Third-party behaviour restated (crates absent from /root/reference, only
 * semver ranges in Cargo.toml:18-28, no Cargo.lock):
 *   regex = "1.5": the format regex is a concatenation of fixed-length
 *     pieces -- "(?P<name>.{n})", upper-case literals, "[AGCT]{n}"
 *     (src/info.rs:263-267, 291-294, 298) -- so leftmost-first search is
 *     "the smallest offset at which every piece matches"; '.' matches any
 *     character except '\n'.  ASCII input is assumed.
 *   ahash = "0.8": set membership / map insertion only (src/parse.rs:457,
 *     489; src/info.rs:745-800); iteration order never changes a result
 *     (fix_error is order independent), so any hash table does.
 */
#include "oracle.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

/* ------------------------------------------------------------------ */
/* small string-keyed hash map (stands in for ahash HashMap / AHashSet) */
/* ------------------------------------------------------------------ */
typedef struct {
  char *key; /* NUL-terminated copy; NULL = empty slot */
  size_t klen;
  uint64_t val;
} smap_ent;

typedef struct {
  smap_ent *e;
  size_t cap; /* power of two */
  size_t n;
} smap;

static uint64_t fnv1a(const char *s, size_t n) {
  uint64_t h = 1469598103934665603ULL;
  for (size_t i = 0; i < n; i++) {
    h ^= (unsigned char)s[i];
    h *= 1099511628211ULL;
  }
  return h ^ (h >> 29);
}

static void smap_init(smap *m) {
  m->cap = 16;
  m->n = 0;
  m->e = (smap_ent *)calloc(m->cap, sizeof(smap_ent));
}

static void smap_free_keys(smap *m) {
  if (!m->e) return;
  for (size_t i = 0; i < m->cap; i++) free(m->e[i].key);
  free(m->e);
  m->e = NULL;
  m->cap = m->n = 0;
}

static smap_ent *smap_find(const smap *m, const char *k, size_t klen) {
  size_t mask = m->cap - 1;
  size_t i = (size_t)fnv1a(k, klen) & mask;
  for (;;) {
    smap_ent *e = &m->e[i];
    if (!e->key) return NULL;
    if (e->klen == klen && memcmp(e->key, k, klen) == 0) return e;
    i = (i + 1) & mask;
  }
}

static void smap_grow(smap *m);

/* returns the entry for k, inserting (val=0) when absent; *inserted says which */
static smap_ent *smap_entry(smap *m, const char *k, size_t klen, int *inserted) {
  if ((m->n + 1) * 4 > m->cap * 3) smap_grow(m);
  size_t mask = m->cap - 1;
  size_t i = (size_t)fnv1a(k, klen) & mask;
  for (;;) {
    smap_ent *e = &m->e[i];
    if (!e->key) {
      e->key = (char *)malloc(klen + 1);
      memcpy(e->key, k, klen);
      e->key[klen] = 0;
      e->klen = klen;
      e->val = 0;
      m->n++;
      if (inserted) *inserted = 1;
      return e;
    }
    if (e->klen == klen && memcmp(e->key, k, klen) == 0) {
      if (inserted) *inserted = 0;
      return e;
    }
    i = (i + 1) & mask;
  }
}

static void smap_grow(smap *m) {
  smap old = *m;
  m->cap = old.cap * 2;
  m->e = (smap_ent *)calloc(m->cap, sizeof(smap_ent));
  size_t mask = m->cap - 1;
  for (size_t j = 0; j < old.cap; j++) {
    if (!old.e[j].key) continue;
    size_t i = (size_t)fnv1a(old.e[j].key, old.e[j].klen) & mask;
    while (m->e[i].key) i = (i + 1) & mask;
    m->e[i] = old.e[j];
  }
  free(old.e);
}

/* ------------------------------------------------------------------ */
/* compiled scheme: SequenceFormat, src/info.rs:176-187                 */
/* ------------------------------------------------------------------ */
enum { PIECE_ANY = 0, PIECE_ACGT = 1, PIECE_LIT = 2 };
enum { GROUP_SAMPLE = 0, GROUP_BARCODE = 1, GROUP_RANDOM = 2 };

typedef struct {
  int kind;
  uint32_t n; /* repeat count for ANY / ACGT; 1 for LIT */
  char lit;
  int group; /* index into groups[] for ANY pieces, -1 otherwise */
} piece;

typedef struct {
  int type;        /* GROUP_* */
  uint32_t number; /* barcode number (1-based) for GROUP_BARCODE */
  uint32_t off;    /* byte offset of the capture inside a match */
  uint32_t len;
} group;

#define ORC_MAX_BARCODES 16

struct orc_ctx {
  /* SequenceFormat */
  char *format_string;
  char *regions_string;
  char *regex_string;
  uint32_t length;
  uint32_t regex_len; /* bytes one match spans */
  uint32_t constant_region_length;
  uint32_t barcode_num;
  uint32_t barcode_lengths[ORC_MAX_BARCODES];
  int32_t sample_length; /* -1 = None */
  int random_barcode;
  int sample_barcode;
  piece *pieces;
  uint32_t n_pieces;
  group *groups;
  uint32_t n_groups;
  int sample_group;                    /* index into groups or -1 */
  int random_group;                    /* index into groups or -1 */
  int barcode_group[ORC_MAX_BARCODES]; /* group index of barcode{i+1} */

  /* BarcodeConversions, src/info.rs:338-343 */
  smap samples_barcode_hash;                    /* seq -> id (char*) */
  smap counted_barcodes_hash[ORC_MAX_BARCODES]; /* per position: seq -> id */
  int counted_loaded;                           /* counted_barcodes_hash non-empty */

  /* MaxSeqErrors, src/info.rs:461-472 */
  int opt_sample_errors, opt_barcode_errors, opt_constant_errors; /* -1 = None */
  uint16_t max_constant, max_sample, max_barcode[ORC_MAX_BARCODES];
  float min_quality;

  /* SequenceErrors, src/info.rs:16-23 (u32 in the reference; kept wide here) */
  uint64_t counters[ORC_NCOUNTERS];
  uint64_t undefined_reads;

  /* Results, src/info.rs:669-674 */
  smap results; /* sample -> (smap*) tuple -> count | (smap*) random set */
  int results_init;
  int sample_conversion_omited;

  /* flattened rows for the accessors */
  struct row {
    const char *sample, *tuple;
    uint64_t count;
  } *rows;
  uint64_t n_rows;
  int rows_valid;

  /* scratch */
  char *seqbuf;
  size_t seqcap;

  /* orc_run_reference_threads: this context is one worker's clone of the static inputs (src/main.rs:95-102); its
   * add_count goes to the job's one shared Results under the job's one mutex (Arc<Mutex<Results>>, src/parse.rs:60) */
  struct orc_ctx *shared_results;
  void *results_mutex; /* pthread_mutex_t* */
};

static char *xstrndup(const char *s, size_t n) {
  char *r = (char *)malloc(n + 1);
  memcpy(r, s, n);
  r[n] = 0;
  return r;
}

typedef struct {
  char *p;
  size_t n, cap;
} sbuf;
static void sb_push(sbuf *b, const char *s, size_t n) {
  if (b->n + n + 1 > b->cap) {
    b->cap = (b->n + n + 1) * 2 + 16;
    b->p = (char *)realloc(b->p, b->cap);
  }
  memcpy(b->p + b->n, s, n);
  b->n += n;
  b->p[b->n] = 0;
}
static void sb_pushc(sbuf *b, char c) { sb_push(b, &c, 1); }

/* match one token of (?i)(\{\d+\})|(\[\d+\])|(\(\d+\))|N+|[ATGC]+ at s[i..n);
 * returns its length or 0 (src/info.rs:232) */
static size_t scheme_token(const char *s, size_t i, size_t n) {
  char c = s[i];
  if (c == '{' || c == '[' || c == '(') {
    char close = c == '{' ? '}' : (c == '[' ? ']' : ')');
    size_t j = i + 1;
    while (j < n && s[j] >= '0' && s[j] <= '9') j++;
    if (j > i + 1 && j < n && s[j] == close) return j + 1 - i;
    return 0;
  }
  if (c == 'N' || c == 'n') {
    size_t j = i;
    while (j < n && (s[j] == 'N' || s[j] == 'n')) j++;
    return j - i;
  }
  {
    size_t j = i;
    while (j < n) {
      char u = (char)toupper((unsigned char)s[j]);
      if (u == 'A' || u == 'T' || u == 'G' || u == 'C')
        j++;
      else
        break;
    }
    return j - i;
  }
}

static void add_piece(orc_ctx *c, int kind, uint32_t n, char lit, int grp) {
  c->pieces = (piece *)realloc(c->pieces, (c->n_pieces + 1) * sizeof(piece));
  piece *p = &c->pieces[c->n_pieces++];
  p->kind = kind;
  p->n = n;
  p->lit = lit;
  p->group = grp;
}

/* SequenceFormat::parse_format_file, src/info.rs:215-310 */
orc_ctx *orc_new(const char *text, size_t len, char *err, size_t errlen) {
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->sample_length = -1;
  c->sample_group = c->random_group = -1;
  c->opt_sample_errors = c->opt_barcode_errors = c->opt_constant_errors = -1;
  smap_init(&c->samples_barcode_hash);
  for (int i = 0; i < ORC_MAX_BARCODES; i++) smap_init(&c->counted_barcodes_hash[i]);
  smap_init(&c->results);

  /* .lines().filter(|line| !line.starts_with('#')).collect::<String>()  info.rs:218-222 */
  sbuf data = {0};
  sb_push(&data, "", 0);
  size_t i = 0;
  while (i < len) {
    size_t j = i;
    while (j < len && text[j] != '\n') j++;
    size_t e = j;
    if (e > i && text[e - 1] == '\r') e--; /* str::lines strips "\r\n" */
    if (!(e > i && text[i] == '#')) sb_push(&data, text + i, e - i);
    i = j + 1;
  }

  sbuf fmt = {0}, reg = {0}, rx = {0};
  sb_push(&fmt, "", 0);
  sb_push(&reg, "", 0);
  sb_push(&rx, "", 0);
  uint32_t off = 0; /* bytes matched so far by the regex */
  size_t pos = 0;
  while (pos < data.n) {
    size_t tl = scheme_token(data.p, pos, data.n);
    if (tl == 0) {
      pos++;
      continue;
    }
    const char *g = data.p + pos;
    int gtype = -1;
    /* info.rs:240-249 */
    if (memchr(g, '[', tl))
      gtype = GROUP_SAMPLE;
    else if (memchr(g, '{', tl))
      gtype = GROUP_BARCODE;
    else if (memchr(g, '(', tl))
      gtype = GROUP_RANDOM;
    if (gtype >= 0) {
      uint32_t digits = (uint32_t)strtoul(g + 1, NULL, 10); /* info.rs:252-259 */
      char name[32];
      if (gtype == GROUP_SAMPLE) {
        if (c->sample_barcode) {
          if (err) snprintf(err, errlen, "regex: duplicate capture group name 'sample'");
          free(data.p), free(fmt.p), free(reg.p), free(rx.p);
          orc_free(c);
          return NULL;
        }
        c->sample_barcode = 1;
        snprintf(name, sizeof name, "sample");
      } else if (gtype == GROUP_BARCODE) {
        c->barcode_num++;
        if (c->barcode_num > ORC_MAX_BARCODES) {
          if (err) snprintf(err, errlen, "oracle: more than %d counted barcodes", ORC_MAX_BARCODES);
          free(data.p), free(fmt.p), free(reg.p), free(rx.p);
          orc_free(c);
          return NULL;
        }
        snprintf(name, sizeof name, "barcode%u", c->barcode_num);
      } else {
        if (c->random_barcode) {
          if (err) snprintf(err, errlen, "regex: duplicate capture group name 'random'");
          free(data.p), free(fmt.p), free(reg.p), free(rx.p);
          orc_free(c);
          return NULL;
        }
        c->random_barcode = 1;
        snprintf(name, sizeof name, "random");
      }
      /* info.rs:263-267 */
      char cap[64];
      snprintf(cap, sizeof cap, "(?P<%s>.{%u})", name, digits);
      sb_push(&rx, cap, strlen(cap));
      c->groups = (group *)realloc(c->groups, (c->n_groups + 1) * sizeof(group));
      group *gr = &c->groups[c->n_groups];
      gr->type = gtype;
      gr->number = gtype == GROUP_BARCODE ? c->barcode_num : 0;
      gr->off = off;
      gr->len = digits;
      add_piece(c, PIECE_ANY, digits, 0, (int)c->n_groups);
      char push_char;
      if (gtype == GROUP_SAMPLE) { /* info.rs:272-280 */
        c->sample_length = (int32_t)digits;
        c->sample_group = (int)c->n_groups;
        push_char = 'S';
      } else if (gtype == GROUP_BARCODE) {
        c->barcode_lengths[c->barcode_num - 1] = digits;
        c->barcode_group[c->barcode_num - 1] = (int)c->n_groups;
        push_char = 'B';
      } else {
        c->random_group = (int)c->n_groups;
        push_char = 'R';
      }
      c->n_groups++;
      for (uint32_t k = 0; k < digits; k++) { /* info.rs:283-286 */
        sb_pushc(&reg, push_char);
        sb_pushc(&fmt, 'N');
      }
      off += digits;
    } else if (memchr(g, 'N', tl)) {
      /* info.rs:287-295: upper-case 'N's only are counted; the whole token
       * goes into format_string; nothing is pushed onto regions_string */
      uint32_t num_of_ns = 0;
      for (size_t k = 0; k < tl; k++)
        if (g[k] == 'N') num_of_ns++;
      char ng[32];
      snprintf(ng, sizeof ng, "[AGCT]{%u}", num_of_ns);
      sb_push(&rx, ng, strlen(ng));
      add_piece(c, PIECE_ACGT, num_of_ns, 0, -1);
      sb_push(&fmt, g, tl);
      off += num_of_ns;
    } else {
      /* info.rs:296-305: constant region; regex gets the upper-cased token,
       * format_string the token as written */
      for (size_t k = 0; k < tl; k++) {
        char u = (char)toupper((unsigned char)g[k]);
        sb_pushc(&rx, u);
        add_piece(c, PIECE_LIT, 1, u, -1);
        sb_pushc(&reg, 'C');
      }
      sb_push(&fmt, g, tl);
      c->constant_region_length += (uint32_t)tl;
      off += (uint32_t)tl;
    }
    pos += tl;
  }
  c->format_string = fmt.p;
  c->regions_string = reg.p;
  c->regex_string = rx.p;
  c->length = (uint32_t)fmt.n; /* info.rs:307 */
  c->regex_len = off;
  free(data.p);
  return c;
}

static void free_results(orc_ctx *c) {
  if (c->results.e) {
    for (size_t i = 0; i < c->results.cap; i++) {
      if (!c->results.e[i].key) continue;
      smap *inner = (smap *)(uintptr_t)c->results.e[i].val;
      if (!inner) continue;
      if (c->random_barcode) {
        for (size_t j = 0; j < inner->cap; j++) {
          if (!inner->e[j].key) continue;
          smap *set = (smap *)(uintptr_t)inner->e[j].val;
          if (set) {
            smap_free_keys(set);
            free(set);
          }
        }
      }
      smap_free_keys(inner);
      free(inner);
    }
  }
  smap_free_keys(&c->results);
  free(c->rows);
  c->rows = NULL;
  c->n_rows = 0;
  c->rows_valid = 0;
}

static void free_idmap(smap *m) {
  if (!m->e) return;
  for (size_t i = 0; i < m->cap; i++)
    if (m->e[i].key) free((void *)(uintptr_t)m->e[i].val);
  smap_free_keys(m);
}

void orc_free(orc_ctx *c) {
  if (!c) return;
  free_results(c);
  free_idmap(&c->samples_barcode_hash);
  for (int i = 0; i < ORC_MAX_BARCODES; i++) free_idmap(&c->counted_barcodes_hash[i]);
  free(c->format_string);
  free(c->regions_string);
  free(c->regex_string);
  free(c->pieces);
  free(c->groups);
  free(c->seqbuf);
  free(c);
}

const char *orc_format_string(const orc_ctx *c) { return c->format_string; }
const char *orc_regions_string(const orc_ctx *c) { return c->regions_string; }
const char *orc_regex_string(const orc_ctx *c) { return c->regex_string; }
uint32_t orc_length(const orc_ctx *c) { return c->length; }
uint32_t orc_constant_region_length(const orc_ctx *c) { return c->constant_region_length; }
uint32_t orc_barcode_num(const orc_ctx *c) { return c->barcode_num; }
uint32_t orc_barcode_length(const orc_ctx *c, uint32_t i) { return c->barcode_lengths[i]; }
int32_t orc_sample_length(const orc_ctx *c) { return c->sample_length; }
int orc_has_random(const orc_ctx *c) { return c->random_barcode; }
int orc_has_sample(const orc_ctx *c) { return c->sample_barcode; }

/* ------------------------------------------------------------------ */
/* CSV loaders: BarcodeConversions, src/info.rs:364-456                 */
/* ------------------------------------------------------------------ */
static void idmap_insert(smap *m, const char *seq, size_t sl, const char *id, size_t il) {
  smap_ent *e = smap_entry(m, seq, sl, NULL);
  free((void *)(uintptr_t)e->val); /* HashMap::insert overwrites: last ID wins */
  e->val = (uint64_t)(uintptr_t)xstrndup(id, il);
}

/* split one line on ',' and return up to `want` fields; returns field count
 * seen (capped at want) */
static int split_fields(const char *l, size_t n, int want, const char **f, size_t *fl) {
  int k = 0;
  size_t s = 0;
  for (size_t i = 0; i <= n && k < want; i++) {
    if (i == n || l[i] == ',') {
      f[k] = l + s;
      fl[k] = i - s;
      k++;
      s = i + 1;
    }
  }
  return k;
}

/* iterate str::lines(): split on '\n', strip one trailing '\r' */
static int next_line(const char *t, size_t len, size_t *pos, const char **l, size_t *ll) {
  if (*pos >= len) return 0;
  size_t i = *pos, j = i;
  while (j < len && t[j] != '\n') j++;
  size_t e = j;
  if (e > i && t[e - 1] == '\r') e--;
  *l = t + i;
  *ll = e - i;
  *pos = j + 1;
  return 1;
}

/* src/info.rs:364-381 */
int orc_load_sample_csv(orc_ctx *c, const char *text, size_t len) {
  size_t pos = 0;
  const char *l;
  size_t ll;
  int first = 1;
  while (next_line(text, len, &pos, &l, &ll)) {
    if (first) { /* .skip(1) */
      first = 0;
      continue;
    }
    const char *f[2];
    size_t fl[2];
    int k = split_fields(l, ll, 2, f, fl);
    if (k == 2)
      idmap_insert(&c->samples_barcode_hash, f[0], fl[0], f[1], fl[1]);
    else /* collect_tuple() == None -> ("","")  info.rs:374-375 */
      idmap_insert(&c->samples_barcode_hash, "", 0, "", 0);
  }
  return 0;
}

/* src/info.rs:390-433 */
int orc_load_counted_csv(orc_ctx *c, const char *text, size_t len, char *err, size_t errlen) {
  size_t pos = 0;
  const char *l;
  size_t ll;
  int first = 1;
  uint8_t contained[ORC_MAX_BARCODES] = {0};
  c->counted_loaded = 1; /* info.rs:408-410 pushes barcode_num maps */
  while (next_line(text, len, &pos, &l, &ll)) {
    if (first) {
      first = 0;
      continue;
    }
    const char *f[3];
    size_t fl[3];
    static const char empty[1] = "";
    int k = split_fields(l, ll, 3, f, fl);
    if (k != 3) {
      f[0] = f[1] = f[2] = empty;
      fl[0] = fl[1] = fl[2] = 0;
    }
    /* barcode_num.parse::<usize>()? - 1   info.rs:413-416 */
    char nb[24];
    if (fl[2] == 0 || fl[2] >= sizeof nb) {
      if (err)
        snprintf(err, errlen, "Third column of barcode file contains something other than an integer: %.*s", (int)fl[2],
                 f[2]);
      return -1;
    }
    memcpy(nb, f[2], fl[2]);
    nb[fl[2]] = 0;
    /* usize::from_str: optional '+', then ASCII digits only */
    const char *dp = nb[0] == '+' ? nb + 1 : nb;
    int digits_ok = *dp != 0;
    for (const char *q = dp; *q; q++)
      if (*q < '0' || *q > '9') digits_ok = 0;
    if (!digits_ok) {
      if (err) snprintf(err, errlen, "Third column of barcode file contains something other than an integer: %s", nb);
      return -1;
    }
    unsigned long v = strtoul(dp, NULL, 10);
    if (v == 0 || v > c->barcode_num) {
      /* 0 - 1 underflows / index out of bounds: the reference panics */
      if (err) snprintf(err, errlen, "panic: barcode number %lu out of range 1..=%u", v, c->barcode_num);
      return -2;
    }
    contained[v - 1] = 1;
    idmap_insert(&c->counted_barcodes_hash[v - 1], f[0], fl[0], f[1], fl[1]);
  }
  for (uint32_t x = 0; x < c->barcode_num; x++) { /* info.rs:420-431 */
    if (!contained[x]) {
      if (err) snprintf(err, errlen, "Barcode conversion file missing barcode numers [%u..] in the third column", x);
      return -3;
    }
  }
  return 0;
}

int orc_add_sample(orc_ctx *c, const char *seq, const char *id) {
  idmap_insert(&c->samples_barcode_hash, seq, strlen(seq), id, strlen(id));
  return 0;
}

int orc_add_counted(orc_ctx *c, uint32_t bi, const char *seq, const char *id) {
  if (bi >= c->barcode_num) return -1;
  c->counted_loaded = 1;
  idmap_insert(&c->counted_barcodes_hash[bi], seq, strlen(seq), id, strlen(id));
  return 0;
}

const char *orc_sample_id(const orc_ctx *c, const char *seq) {
  smap_ent *e = smap_find(&c->samples_barcode_hash, seq, strlen(seq));
  return e ? (const char *)(uintptr_t)e->val : NULL;
}
const char *orc_counted_id(const orc_ctx *c, uint32_t bi, const char *seq) {
  if (bi >= ORC_MAX_BARCODES) return NULL;
  smap_ent *e = smap_find(&c->counted_barcodes_hash[bi], seq, strlen(seq));
  return e ? (const char *)(uintptr_t)e->val : NULL;
}

/* ------------------------------------------------------------------ */
/* MaxSeqErrors::new, src/info.rs:490-543                               */
/* ------------------------------------------------------------------ */
void orc_max_seq_errors(int sample_errors, int sample_size, int barcode_errors, const uint16_t *barcode_sizes,
                        uint32_t n_barcodes, int constant_errors, uint16_t constant_region_size, uint16_t *out) {
  uint16_t max_sample_errors;
  if (sample_size >= 0) { /* info.rs:503-513 */
    if (sample_errors >= 0)
      max_sample_errors = (uint16_t)sample_errors;
    else
      max_sample_errors = (uint16_t)(sample_size / 5);
  } else {
    max_sample_errors = 0;
  }
  for (uint32_t i = 0; i < n_barcodes; i++) { /* info.rs:517-523 */
    if (barcode_errors >= 0)
      out[2 + i] = (uint16_t)barcode_errors;
    else
      out[2 + i] = (uint16_t)(barcode_sizes[i] / 5);
  }
  uint16_t max_constant_errors; /* info.rs:527-532 */
  if (constant_errors >= 0)
    max_constant_errors = (uint16_t)constant_errors;
  else
    max_constant_errors = (uint16_t)(constant_region_size / 5);
  out[0] = max_constant_errors;
  out[1] = max_sample_errors;
}

static void recompute_budgets(orc_ctx *c) {
  /* main.rs:55-63 wires SequenceFormat fields into MaxSeqErrors::new */
  uint16_t sizes[ORC_MAX_BARCODES], out[2 + ORC_MAX_BARCODES];
  for (uint32_t i = 0; i < c->barcode_num; i++) sizes[i] = (uint16_t)c->barcode_lengths[i];
  orc_max_seq_errors(c->opt_sample_errors, c->sample_length, c->opt_barcode_errors, sizes, c->barcode_num,
                     c->opt_constant_errors, (uint16_t)c->constant_region_length, out);
  c->max_constant = out[0];
  c->max_sample = out[1];
  for (uint32_t i = 0; i < c->barcode_num; i++) c->max_barcode[i] = out[2 + i];
}

void orc_set_max_errors(orc_ctx *c, int sample_errors, int barcode_errors, int constant_errors) {
  c->opt_sample_errors = sample_errors;
  c->opt_barcode_errors = barcode_errors;
  c->opt_constant_errors = constant_errors;
  recompute_budgets(c);
}
void orc_set_min_quality(orc_ctx *c, float q) { c->min_quality = q; }
uint32_t orc_max_constant_errors(const orc_ctx *c) { return c->max_constant; }
uint32_t orc_max_sample_errors(const orc_ctx *c) { return c->max_sample; }
uint32_t orc_max_barcode_errors(const orc_ctx *c, uint32_t i) { return c->max_barcode[i]; }

/* ------------------------------------------------------------------ */
/* Results::new, src/info.rs:678-732                                    */
/* ------------------------------------------------------------------ */
static smap *new_smap(void) {
  smap *m = (smap *)malloc(sizeof(smap));
  smap_init(m);
  return m;
}

void orc_begin(orc_ctx *c) {
  free_results(c);
  smap_init(&c->results);
  recompute_budgets(c);
  memset(c->counters, 0, sizeof c->counters);
  c->undefined_reads = 0;
  c->sample_conversion_omited = 0;
  if (c->samples_barcode_hash.n != 0) { /* info.rs:698-709 */
    for (size_t i = 0; i < c->samples_barcode_hash.cap; i++) {
      smap_ent *s = &c->samples_barcode_hash.e[i];
      if (!s->key) continue;
      smap_ent *e = smap_entry(&c->results, s->key, s->klen, NULL);
      if (!e->val) e->val = (uint64_t)(uintptr_t)new_smap();
    }
  } else if (!c->sample_barcode) { /* info.rs:710-719 */
    smap_ent *e = smap_entry(&c->results, "barcode", 7, NULL);
    e->val = (uint64_t)(uintptr_t)new_smap();
  } else { /* info.rs:720-724 */
    c->sample_conversion_omited = 1;
  }
  c->results_init = 1;
}

/* Results::add_count, src/info.rs:735-808 */
static int add_count(orc_ctx *c, const char *sample, size_t sl, const char *random, size_t rl, int has_random_arg,
                     const char *tuple, size_t tl) {
  c->rows_valid = 0;
  if (c->sample_conversion_omited) { /* info.rs:742-757 */
    int ins;
    smap_ent *e = smap_entry(&c->results, sample, sl, &ins);
    if (ins) e->val = (uint64_t)(uintptr_t)new_smap();
  }
  if (!c->random_barcode) { /* info.rs:761-767 */
    smap_ent *s = smap_find(&c->results, sample, sl);
    if (s) {
      smap *inner = (smap *)(uintptr_t)s->val;
      smap_ent *e = smap_entry(inner, tuple, tl, NULL);
      e->val += 1;
    }
    /* else: unwrap_or(&mut clone) -> the increment lands in a temporary */
    return 1;
  }
  /* info.rs:770-802 */
  if (!has_random_arg) { /* random_barcode.unwrap_or(&"".to_string()) */
    random = "";
    rl = 0;
  }
  smap_ent *s = (sl == 0) ? smap_find(&c->results, "barcode", 7) : smap_find(&c->results, sample, sl);
  if (s) {
    smap *inner = (smap *)(uintptr_t)s->val;
    int ins;
    smap_ent *e = smap_entry(inner, tuple, tl, &ins);
    if (ins) { /* Entry::Vacant: info.rs:780-785 */
      smap *set = new_smap();
      smap_entry(set, random, rl, NULL);
      e->val = (uint64_t)(uintptr_t)set;
    } else { /* info.rs:786-791 */
      smap *set = (smap *)(uintptr_t)e->val;
      int ins2;
      smap_entry(set, random, rl, &ins2);
      return ins2;
    }
  } else { /* info.rs:792-801 */
    smap *set = new_smap();
    smap_entry(set, random, rl, NULL);
    smap *inner = new_smap();
    smap_ent *e = smap_entry(inner, tuple, tl, NULL);
    e->val = (uint64_t)(uintptr_t)set;
    smap_ent *ns = smap_entry(&c->results, sample, sl, NULL);
    ns->val = (uint64_t)(uintptr_t)inner;
  }
  return 1;
}

/* ------------------------------------------------------------------ */
/* fix_error, src/parse.rs:553-593                                      */
/* ------------------------------------------------------------------ */
typedef struct {
  const char *p;
  size_t n;
} sview;

static int64_t fix_error_views(const char *mismatch_seq, size_t mlen, const sview *possible, uint64_t n,
                               uint16_t mismatches_allowed) {
  int64_t best_match = -1;                                            /* parse.rs:557 */
  uint32_t best_mismatch_count = (uint32_t)mismatches_allowed + 1;    /* parse.rs:558 */
  int keep = 1;                                                       /* parse.rs:559 */
  for (uint64_t t = 0; t < n; t++) {                                  /* parse.rs:562 */
    uint32_t mismatches = 0;                                          /* parse.rs:564 */
    size_t z = possible[t].n < mlen ? possible[t].n : mlen;           /* zip: parse.rs:568 */
    for (size_t k = 0; k < z; k++) {
      char possible_char = possible[t].p[k], current_char = mismatch_seq[k];
      if (possible_char != current_char && current_char != 'N' && possible_char != 'N') mismatches++; /* :569-571 */
      if (mismatches > best_mismatch_count) break;                    /* parse.rs:572-574 */
    }
    if (mismatches == best_mismatch_count) keep = 0;                  /* parse.rs:577-579 */
    if (mismatches < best_mismatch_count) {                           /* parse.rs:581-585 */
      keep = 1;
      best_mismatch_count = mismatches;
      best_match = (int64_t)t;
    }
  }
  if (keep && best_match >= 0) return best_match;                     /* parse.rs:588-592 */
  return -1;
}

int64_t orc_fix_error(const char *mismatch_seq, const char *const *possible, uint64_t n, uint16_t mismatches) {
  sview *v = (sview *)malloc((n ? n : 1) * sizeof(sview));
  for (uint64_t i = 0; i < n; i++) {
    v[i].p = possible[i];
    v[i].n = strlen(possible[i]);
  }
  int64_t r = fix_error_views(mismatch_seq, strlen(mismatch_seq), v, n, mismatches);
  free(v);
  return r;
}

/* fix_error over the keys of a set; returns the matching entry or NULL */
static const smap_ent *fix_error_set(const char *q, size_t ql, const smap *set, uint16_t max) {
  sview *v = (sview *)malloc((set->n ? set->n : 1) * sizeof(sview));
  const smap_ent **ents = (const smap_ent **)malloc((set->n ? set->n : 1) * sizeof(smap_ent *));
  uint64_t n = 0;
  for (size_t i = 0; i < set->cap; i++) {
    if (!set->e[i].key) continue;
    v[n].p = set->e[i].key;
    v[n].n = set->e[i].klen;
    ents[n] = &set->e[i];
    n++;
  }
  int64_t r = fix_error_views(q, ql, v, n, max);
  const smap_ent *res = r >= 0 ? ents[r] : NULL;
  free(v);
  free(ents);
  return res;
}

/* ------------------------------------------------------------------ */
/* the format regex, restated (see header comment)                      */
/* ------------------------------------------------------------------ */
static int regex_match_at(const orc_ctx *c, const char *s, size_t n, size_t o) {
  if (o + c->regex_len > n) return 0;
  size_t p = o;
  for (uint32_t i = 0; i < c->n_pieces; i++) {
    const piece *pc = &c->pieces[i];
    switch (pc->kind) {
      case PIECE_LIT:
        if (s[p] != pc->lit) return 0;
        p++;
        break;
      case PIECE_ACGT:
        for (uint32_t k = 0; k < pc->n; k++, p++)
          if (!(s[p] == 'A' || s[p] == 'G' || s[p] == 'C' || s[p] == 'T')) return 0;
        break;
      default:
        for (uint32_t k = 0; k < pc->n; k++, p++)
          if (s[p] == '\n') return 0;
        break;
    }
  }
  return 1;
}

/* Regex::find / captures / is_match: leftmost match; returns offset or -1 */
static int64_t regex_find(const orc_ctx *c, const char *s, size_t n) {
  if (c->regex_len > n) return -1;
  for (size_t o = 0; o + c->regex_len <= n; o++)
    if (regex_match_at(c, s, n, o)) return (int64_t)o;
  return -1;
}

/* ------------------------------------------------------------------ */
/* RawSequenceRead::low_quality, src/parse.rs:323-375                   */
/* ------------------------------------------------------------------ */
static int low_quality(const char *qual, size_t ql, float min_average, const char *regions, size_t start) {
  /* scores: Vec<f32>; only its running sum and length matter. f32 adds of
   * integers < 2^24 are exact, so summing as we go equals iter().sum() */
  float sum = 0.0f;
  uint32_t cnt = 0;
  char previous_type = '\0'; /* parse.rs:338 */
  size_t rl = strlen(regions);
  for (size_t k = 0; start + k < ql && k < rl; k++) { /* skip(start).zip(regions)  parse.rs:340-345 */
    uint8_t score = (uint8_t)((uint8_t)qual[start + k] - 33); /* parse.rs:326, release-mode wrap */
    char seq_type = regions[k];
    if (seq_type != previous_type) { /* parse.rs:348 */
      if (cnt != 0) {                /* parse.rs:350 */
        float average_score = sum / (float)cnt; /* parse.rs:352-353 */
        if (average_score < min_average) return 1; /* parse.rs:354-356 */
        sum = 0.0f;
        cnt = 0; /* parse.rs:358 */
      }
      previous_type = seq_type;                       /* parse.rs:361 */
      if (seq_type != 'C') {                           /* parse.rs:363-365 */
        sum = (float)score;
        cnt = 1;
      }
    } else if (seq_type != 'C') { /* parse.rs:368-370 */
      sum += (float)score;
      cnt++;
    }
  }
  return 0; /* parse.rs:374 */
}

/* ------------------------------------------------------------------ */
/* SequenceParser::parse body for one read, src/parse.rs:53-163, 430-529 */
/* ------------------------------------------------------------------ */
int orc_process_read(orc_ctx *c, const char *seq_in, size_t seqlen, const char *qual, size_t quallen) {
  if (!c->results_init) orc_begin(c);
  if (c->seqcap < seqlen + c->length + 2) {
    c->seqcap = (seqlen + c->length + 2) * 2;
    c->seqbuf = (char *)realloc(c->seqbuf, c->seqcap);
  }
  char *seq = c->seqbuf;
  memcpy(seq, seq_in, seqlen);
  size_t n = seqlen;

  /* check_and_fix_consant_region, parse.rs:151-163 */
  if (regex_find(c, seq, n) < 0) {
    /* RawSequenceRead::fix_constant_region, parse.rs:287-313 */
    size_t L = c->length;
    if (n < L) {
      /* parse.rs:291: usize underflow -- panic in debug, runaway loop in
       * release.  Undefined in the reference; we end the read as a
       * constant-region error and count it separately. */
      c->undefined_reads++;
      n = 0;
    } else {
      size_t length_diff = n - L; /* parse.rs:291 */
      sview *windows = (sview *)malloc((length_diff ? length_diff : 1) * sizeof(sview));
      for (size_t index = 0; index < length_diff; index++) { /* parse.rs:295-304 */
        windows[index].p = seq + index;
        windows[index].n = L;
      }
      int64_t best = fix_error_views(c->format_string, L, windows, length_diff, c->max_constant); /* :306 */
      free(windows);
      if (best >= 0) {
        /* insert_barcodes_constant_region, parse.rs:270-283 */
        char *fixed = seq + seqlen + 1;
        for (size_t k = 0; k < L; k++) {
          char old_char = seq[(size_t)best + k], new_char = c->format_string[k];
          fixed[k] = (new_char == 'N') ? old_char : new_char;
        }
        memmove(seq, fixed, L);
        n = L;
      } else {
        n = 0; /* parse.rs:311 */
      }
    }
  }

  /* match_seq, parse.rs:89-148 */
  int64_t m = regex_find(c, seq, n); /* captures: parse.rs:92-95 */
  if (m < 0) {
    c->counters[ORC_CONSTANT_REGION]++; /* parse.rs:145 */
    return ORC_CONSTANT_REGION;
  }
  if (c->min_quality > 0.0f) { /* parse.rs:98-119 */
    size_t start = (size_t)m; /* find().start() */
    if (low_quality(qual, quallen, c->min_quality, c->regions_string, start)) {
      c->counters[ORC_LOW_QUALITY]++; /* parse.rs:111 */
      return ORC_LOW_QUALITY;
    }
  }

  /* SequenceMatchResult::new, parse.rs:439-524 */
  const char *base = seq + m;
  int sample_barcode_error = 0;
  const char *sample_barcode;
  size_t sample_len;
  if (c->sample_group >= 0) { /* parse.rs:451 */
    const group *g = &c->groups[c->sample_group];
    const char *s = base + g->off;
    if (c->samples_barcode_hash.n == 0) { /* sample_seqs.is_empty()  parse.rs:453 */
      sample_barcode = s;
      sample_len = g->len;
    } else if (smap_find(&c->samples_barcode_hash, s, g->len)) { /* parse.rs:457 */
      sample_barcode = s;
      sample_len = g->len;
    } else {
      const smap_ent *fx = fix_error_set(s, g->len, &c->samples_barcode_hash, c->max_sample); /* :461-462 */
      if (fx) {
        sample_barcode = fx->key;
        sample_len = fx->klen;
      } else {
        sample_barcode = "";
        sample_len = 0;
        sample_barcode_error = 1; /* parse.rs:466-467 */
      }
    }
  } else {
    sample_barcode = "barcode"; /* parse.rs:473 */
    sample_len = 7;
  }

  int counted_barcode_error = 0;
  sbuf tuple = {0};
  sb_push(&tuple, "", 0);
  if (!sample_barcode_error) { /* parse.rs:481 */
    for (uint32_t index = 0; index < c->barcode_num; index++) { /* parse.rs:483 */
      const group *g = &c->groups[c->barcode_group[index]];
      const char *b = base + g->off;
      size_t bl = g->len;
      if (c->counted_loaded) { /* !counted_barcode_seqs.is_empty()  parse.rs:487 */
        const smap *set = &c->counted_barcodes_hash[index];
        if (!smap_find(set, b, bl)) { /* parse.rs:489 */
          const smap_ent *fx = fix_error_set(b, bl, set, c->max_barcode[index]); /* :490-494 */
          if (fx) {
            b = fx->key;
            bl = fx->klen;
          } else {
            counted_barcode_error = 1; /* parse.rs:499-500 */
            break;
          }
        }
      }
      if (index) sb_pushc(&tuple, ','); /* barcode_string(): join(",")  parse.rs:527-529 */
      sb_push(&tuple, b, bl);
    }
  }
  if (sample_barcode_error) { /* parse.rs:132-135 */
    free(tuple.p);
    c->counters[ORC_SAMPLE_BARCODE]++;
    return ORC_SAMPLE_BARCODE;
  }
  if (counted_barcode_error) { /* parse.rs:137-140 */
    free(tuple.p);
    c->counters[ORC_BARCODE]++;
    return ORC_BARCODE;
  }
  const char *random = NULL;
  size_t random_len = 0;
  if (c->random_group >= 0) { /* parse.rs:512-516 */
    random = base + c->groups[c->random_group].off;
    random_len = c->groups[c->random_group].len;
  }
  /* parse.rs:58-69 */
  int added;
  if (c->shared_results) { /* self.shared_mut.results.lock().unwrap().add_count(..), parse.rs:60-64 */
    pthread_mutex_lock((pthread_mutex_t *)c->results_mutex);
    added = add_count(c->shared_results, sample_barcode, sample_len, random, random_len, random != NULL, tuple.p, tuple.n);
    pthread_mutex_unlock((pthread_mutex_t *)c->results_mutex);
  } else {
    added = add_count(c, sample_barcode, sample_len, random, random_len, random != NULL, tuple.p, tuple.n);
  }
  free(tuple.p);
  if (added) {
    c->counters[ORC_MATCHED]++;
    return ORC_MATCHED;
  }
  c->counters[ORC_DUPLICATES]++;
  return ORC_DUPLICATES;
}

void orc_process_batch(orc_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint16_t *lens, uint32_t stride,
                       uint32_t read_len, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) {
    size_t l = lens ? lens[i] : read_len;
    const char *s = (const char *)seq + (size_t)i * stride;
    const char *q = qual ? (const char *)qual + (size_t)i * stride : "";
    orc_process_read(c, s, l, q, qual ? l : 0);
  }
}

/* the same, also reporting which counter each read bumped (exhaustive-domain tests compare read by read) */
void orc_process_batch_outcomes(orc_ctx *c, const uint8_t *seq, const uint8_t *qual, const uint16_t *lens,
                                const uint16_t *qlens, uint32_t stride, uint32_t read_len, uint64_t n, uint8_t *outcomes) {
  for (uint64_t i = 0; i < n; i++) {
    size_t l = lens ? lens[i] : read_len;
    size_t ql = qlens ? qlens[i] : l;
    const char *s = (const char *)seq + (size_t)i * stride;
    const char *q = qual ? (const char *)qual + (size_t)i * stride : "";
    outcomes[i] = (uint8_t)orc_process_read(c, s, l, q, qual ? ql : 0);
  }
}

/* ------------------------------------------------------------------ */
/* The reference's own thread structure, for the CPU baseline: one reader  */
/* thread posting packed 4-line records to a mutex-guarded deque with the  */
/* 10,000-record back-pressure spin (src/input.rs:115-148), threads-1     */
/* workers popping from it and busy-spinning while it is empty            */
/* (src/parse.rs:53-86), one mutex-guarded Results (src/parse.rs:60-64).  */
/* ------------------------------------------------------------------ */
typedef struct qnode {
  struct qnode *prev, *next;
  size_t len;
  char rec[];
} qnode;

typedef struct {
  pthread_mutex_t qmu; /* Arc<Mutex<VecDeque<String>>> */
  qnode *front, *back;
  size_t qlen;
  volatile int finished; /* Arc<AtomicBool> */
  pthread_mutex_t rmu; /* Arc<Mutex<Results>> */
  const uint8_t *seq, *qual;
  uint32_t stride, read_len;
  uint64_t n;
} refjob;

static void *ref_reader(void *arg) { /* read_fastq, src/input.rs:24-89 */
  refjob *j = (refjob *)arg;
  for (uint64_t i = 0; i < j->n; i++) {
    /* the record as the reference queues it: "desc\nseq\n+\nqual" (src/input.rs:124-147) */
    size_t rl = j->read_len, len = 2 + rl + 3 + rl;
    qnode *nd = (qnode *)malloc(sizeof(qnode) + len + 1);
    char *p = nd->rec;
    *p++ = '@';
    *p++ = '\n';
    memcpy(p, j->seq + (size_t)i * j->stride, rl);
    p += rl;
    *p++ = '\n';
    *p++ = '+';
    *p++ = '\n';
    if (j->qual) memcpy(p, j->qual + (size_t)i * j->stride, rl); else memset(p, 'I', rl);
    p[rl] = 0;
    nd->len = len;
    for (;;) { /* back-pressure: spin while 10,000 records are queued (src/input.rs:117-122) */
      pthread_mutex_lock(&j->qmu);
      if (j->qlen < 10000) break;
      pthread_mutex_unlock(&j->qmu);
    }
    nd->prev = NULL; /* push_front (src/input.rs:147) */
    nd->next = j->front;
    if (j->front) j->front->prev = nd; else j->back = nd;
    j->front = nd;
    j->qlen++;
    pthread_mutex_unlock(&j->qmu);
  }
  j->finished = 1; /* src/main.rs:88 */
  return NULL;
}

typedef struct {
  refjob *job;
  orc_ctx *ctx;
} refworker;

static void *ref_worker(void *arg) { /* SequenceParser::parse, src/parse.rs:53-76 */
  refworker *w = (refworker *)arg;
  refjob *j = w->job;
  for (;;) {
    pthread_mutex_lock(&j->qmu); /* get_seqeunce: lock, pop_back (src/parse.rs:78-86) */
    qnode *nd = j->back;
    if (nd) {
      j->back = nd->prev;
      if (j->back) j->back->next = NULL; else j->front = NULL;
      j->qlen--;
    }
    pthread_mutex_unlock(&j->qmu);
    if (!nd) {
      if (j->finished) { /* queue empty and the reader is done (src/parse.rs:71-73); re-check under the lock */
        pthread_mutex_lock(&j->qmu);
        int empty = j->back == NULL;
        pthread_mutex_unlock(&j->qmu);
        if (empty) break;
      }
      continue; /* busy spin: the reference has no condvar */
    }
    /* RawSequenceRead::unpack: split on '\n' into the four lines (src/parse.rs:260-267) */
    const char *l1 = memchr(nd->rec, '\n', nd->len);
    const char *s = l1 + 1;
    const char *l2 = memchr(s, '\n', nd->len - (size_t)(s - nd->rec));
    const char *l3 = memchr(l2 + 1, '\n', nd->len - (size_t)(l2 + 1 - nd->rec));
    const char *q = l3 + 1;
    orc_process_read(w->ctx, s, (size_t)(l2 - s), q, nd->len - (size_t)(q - nd->rec));
    free(nd);
  }
  return NULL;
}

/* workers[0 .. n_workers): identically configured contexts (the per-worker clones); shared: one more, whose Results
 * receives every add_count.  Counters end up per worker (sum them).  Returns 0, or -1 when a thread cannot start. */
int orc_run_reference_threads(orc_ctx **workers, uint32_t n_workers, orc_ctx *shared, const uint8_t *seq,
                              const uint8_t *qual, uint32_t stride, uint32_t read_len, uint64_t n) {
  refjob j;
  memset(&j, 0, sizeof j);
  pthread_mutex_init(&j.qmu, NULL);
  pthread_mutex_init(&j.rmu, NULL);
  j.seq = seq;
  j.qual = qual;
  j.stride = stride;
  j.read_len = read_len;
  j.n = n;
  if (!shared->results_init) orc_begin(shared);
  refworker *ws = (refworker *)calloc(n_workers, sizeof(refworker));
  pthread_t *th = (pthread_t *)calloc(n_workers + 1, sizeof(pthread_t));
  int rc = 0;
  uint32_t started = 0;
  for (uint32_t k = 0; k < n_workers; k++) {
    workers[k]->shared_results = shared;
    workers[k]->results_mutex = &j.rmu;
    ws[k].job = &j;
    ws[k].ctx = workers[k];
    if (pthread_create(&th[k], NULL, ref_worker, &ws[k]) != 0) { rc = -1; break; }
    started++;
  }
  if (rc == 0) ref_reader(&j); /* the calling thread is the reader */
  j.finished = 1;
  for (uint32_t k = 0; k < started; k++) pthread_join(th[k], NULL);
  for (uint32_t k = 0; k < n_workers; k++) {
    workers[k]->shared_results = NULL;
    workers[k]->results_mutex = NULL;
  }
  free(ws);
  free(th);
  pthread_mutex_destroy(&j.qmu);
  pthread_mutex_destroy(&j.rmu);
  return rc;
}

void orc_counters(const orc_ctx *c, uint64_t out[ORC_NCOUNTERS]) { memcpy(out, c->counters, sizeof c->counters); }
uint64_t orc_undefined_reads(const orc_ctx *c) { return c->undefined_reads; }

static void build_rows(orc_ctx *c) {
  if (c->rows_valid) return;
  free(c->rows);
  c->rows = NULL;
  c->n_rows = 0;
  uint64_t cap = 0;
  for (size_t i = 0; i < c->results.cap; i++) {
    smap_ent *s = &c->results.e[i];
    if (!s->key) continue;
    smap *inner = (smap *)(uintptr_t)s->val;
    for (size_t j = 0; j < inner->cap; j++) {
      smap_ent *t = &inner->e[j];
      if (!t->key) continue;
      if (c->n_rows == cap) {
        cap = cap ? cap * 2 : 1024;
        c->rows = (struct row *)realloc(c->rows, cap * sizeof(struct row));
      }
      c->rows[c->n_rows].sample = s->key;
      c->rows[c->n_rows].tuple = t->key;
      c->rows[c->n_rows].count = c->random_barcode ? ((smap *)(uintptr_t)t->val)->n : t->val; /* output.rs:265-270 */
      c->n_rows++;
    }
  }
  c->rows_valid = 1;
}

uint64_t orc_result_rows(orc_ctx *c) {
  if (!c->results_init) orc_begin(c);
  build_rows(c);
  return c->n_rows;
}

int orc_result_row(orc_ctx *c, uint64_t i, const char **sample, const char **tuple, uint64_t *count) {
  build_rows(c);
  if (i >= c->n_rows) return -1;
  *sample = c->rows[i].sample;
  *tuple = c->rows[i].tuple;
  *count = c->rows[i].count;
  return 0;
}

uint64_t orc_result_samples(orc_ctx *c) {
  if (!c->results_init) orc_begin(c);
  return c->results.n;
}

const char *orc_result_sample(orc_ctx *c, uint64_t i) {
  uint64_t k = 0;
  for (size_t j = 0; j < c->results.cap; j++) {
    if (!c->results.e[j].key) continue;
    if (k == i) return c->results.e[j].key;
    k++;
  }
  return NULL;
}
