/*
 * oracle_index.h -- data structures of the oracle's own `pandora index` restatement (oracle_index.c), shared with the
 * oracle's VCF-site restatement (oracle_vcf.c).  TEST INFRASTRUCTURE ONLY: nothing under drprg_amd/ includes this.
 */
#ifndef ORACLE_INDEX_H
#define ORACLE_INDEX_H
#include <stdint.h>

uint64_t orc_hash64(uint64_t key, uint64_t mask); /* oracle.c: the k-mer hash in force (orc_set_hash_mode) */

#define NO_OFF 0xFFFFFFFFu /* item.off of a crossed empty node */

typedef struct {
    uint32_t start, end; /* [start, end) in the PRG string */
    uint32_t n_out, cap_out;
    uint32_t* out;
} lnode;

typedef struct {
    uint32_t node, off; /* base `off` of local node `node`, or (empty node, NO_OFF) */
} item;

typedef struct {
    uint32_t* iv; /* n_iv (start, end) pairs; empty nodes crossed appear as (c, c) */
    uint32_t n_iv;
    item* items; /* the k bases and the empty nodes between them */
    uint32_t n_items;
    uint64_t hash;
    uint8_t strand;
    uint8_t walk_confirmed;
    uint32_t id;
} knode;

typedef struct {
    uint32_t from, to; /* indexes into kn[], or SRC / SNK */
} kedge;

#define SRC 0xFFFFFFFEu
#define SNK 0xFFFFFFFDu

typedef struct orc_kgraph {
    char* s;
    uint32_t len;
    lnode* ln;
    uint32_t n_ln, cap_ln;
    int w, k;
    knode* kn;
    uint32_t n_kn, cap_kn;
    uint32_t* table; /* open addressing over kn[] by interval list; 0 = empty, else index + 1 */
    uint32_t table_size;
    kedge* edges;
    uint64_t n_edges, cap_edges;
    uint32_t* order; /* order[id - 1] = index into kn[] */
    uint32_t min_path_len;
    int error; /* 1 = malformed PRG */
} orc_kgraph;


#endif
