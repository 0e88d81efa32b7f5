/*
 * drprg_hip.h -- C ABI of the MI355X-native drprg predict hot path (libdrprg_hip.so).
 *
 * The reference has no FFI for this path: drprg drives the external `pandora` executable through
 * struct Pandora (/root/reference/src/lib.rs:459-698).  Each entry point below names the reference
 * interface it replaces; a Rust host binds them with `extern "C"` (stub in INTEGRATION.md), or keeps
 * using the process boundary through the drop-in executable drprg_amd/bin/pandora (`drprg predict -p`,
 * /root/reference/src/predict.rs:136-144).
 *
 * Conventions: every function returns 0 on success or a negative errno-style code; no exception
 * crosses the ABI; `drprg_hip_last_error` gives the message of the last failure on that context
 * (or, with ctx == NULL, of the last failed open/index call on the calling thread).  A context is
 * not thread-safe.  Buffers passed in stay owned by the caller; nothing returned needs freeing
 * except the context itself (`drprg_hip_close`).  Plain pointers and sizes only.
 */
#ifndef DRPRG_HIP_H
#define DRPRG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct drprg_hip_ctx drprg_hip_ctx;

/* Mapping options = the argv drprg builds for `pandora map/discover`
 * (/root/reference/src/predict.rs:236-245, src/lib.rs:594-618).
 * ZERO-INITIALISE the struct (memset / `= {0}` / Rust `Default`) before setting fields: every field has a default at 0 and new
 * fields are only ever added where 0 keeps the previous behaviour (`binomial` took what was tail padding in round 3:
 * sizeof stayed 48, and a host that filled the struct field by field without zeroing it would pass an indeterminate value).
 * DRPRG_HIP_MAP_OPTS_SIZE is checked by drprg_hip_set_opts_sized, the entry a host built against another header revision
 * should call: a size other than this library's is refused with -EINVAL instead of being read past or short. */
#define DRPRG_HIP_MAP_OPTS_SIZE 48
typedef struct drprg_hip_map_opts {
    int32_t max_diff;          /* --max-diff; <=0: default (250, or 2k+1 with illumina) */
    double error_rate;         /* -e; <=0: default (0.11, or 0.001 with illumina) */
    uint32_t min_cluster_size; /* -c (drprg passes 10) */
    int32_t illumina;          /* -I */
    uint64_t genome_size;      /* -g (drprg passes 4411532) */
    double genotyping_error_rate; /* <=0: 0.01 */
    int32_t kernel;            /* 0 auto (2 when it applies, else 3); 1 direct sketch kernel + generic cluster pipeline (radix sort);
                                * 2 Bloom-prefiltered kernel (k<=15, w<=16, small index); 3 direct sketch kernel, candidate form */
    int32_t binomial;          /* --bin: binomial model of the k-mer coverages (pandora's default, and what drprg runs, is the negative
                                * binomial: the argv of /root/reference/src/lib.rs:594-618 has no --bin) */
} drprg_hip_map_opts;

/* Replaces Pandora::index_with (`pandora index -t T -w W -k K <prg>`, /root/reference/src/lib.rs:479-510):
 * writes <prg>.k<K>.w<W>.idx and <dir>/kmer_prgs/ beside the PRG (/root/reference/src/builder.rs:263-269).
 * Host-only work; needs no GPU. */
int drprg_hip_index(const char* prg_file, int w, int k, int threads);

/* Opens the index that drprg_hip_index wrote (the files `validate_index` requires,
 * /root/reference/src/predict.rs:400-418) and uploads its probe tables to HIP device `device`.
 * device < 0 opens a host-only context: index export and genotyping work, every map call fails with
 * -ENODEV (there is no CPU fallback for the hot path).  Returns NULL on failure. */
drprg_hip_ctx* drprg_hip_open(const char* prg_file, int w, int k, int device);

/* Same, but sketches the PRG in memory instead of reading .idx / kmer_prgs (no files needed or written). */
drprg_hip_ctx* drprg_hip_open_prg(const char* prg_file, int w, int k, int device, int threads);

/* One context over several HIP devices of a node (the reference is single-process): the index tables
 * are replicated on every device, drprg_hip_map_fastx shards the reads by ingest block over the devices and sums their coverage
 * vectors into devices[0] before it returns (unsigned integer sums: the result is the same whatever device mapped what), so
 * coverage / counters / genotype then see the whole sample.  Every other call (map_host, map_device, device_coverage) uses
 * devices[0].  from_files != 0: like drprg_hip_open, else like drprg_hip_open_prg.  A device may be listed more than once
 * (two concurrent launch sequences on one GPU: what the single-GPU test uses). */
drprg_hip_ctx* drprg_hip_open_multi(const char* prg_file, int w, int k, const int* devices, int ndev, int from_files);

/* ---- the one collective of the path: the sum of the per-GPU coverage vectors (BASELINE north_star "a single
 * RCCL reduce over xGMI").  The reference is single-process and has no counterpart; what these calls feed is the genotyping
 * that follows `pandora map`'s read loop (/root/reference/src/lib.rs:580-642).  RCCL is bound at run time (librccl.so.1):
 * a host that never reduces across GPUs does not need the library.
 *
 * Layout A, one process over several GPUs (drprg_hip_open_multi): drprg_hip_reduce sums the devices' vectors into devices[0]
 * on the devices -- one ncclReduce(sum, u32) over a communicator of the context's devices, or, when a device is listed twice
 * / RCCL is absent / DRPRG_HIP_NO_RCCL=1, a peer copy + add kernel per device -- clears the other devices and folds their
 * counters.  drprg_hip_map_fastx calls it before it returns.  drprg_hip_reduce_info: which of the two ran last. */
int drprg_hip_reduce(drprg_hip_ctx* ctx);
int drprg_hip_reduce_info(const drprg_hip_ctx* ctx, char* out, size_t cap);
/* Layout B, one process per GPU (what `north_star` describes and bench.py --gpus N runs): rank 0 makes an id
 * (drprg_hip_comm_unique_id, 128 bytes) and hands it to the other ranks by whatever channel the host has; every rank calls
 * drprg_hip_comm_init_rank(&comm, nranks, id, rank, device) and, after its last batch, drprg_hip_allreduce(ctx, comm, ...):
 * ONE in-place ncclAllReduce(sum, u32) over the vector [coverage | reads per PRG] (n_covg + n_prgs words of
 * drprg_hip_coverage_size) on `hip_stream` (NULL: the context's stream), asynchronous on that stream -- every rank then holds the
 * sample's vectors and rank 0 genotypes.  d_covg NULL: the context's own accumulators (which are laid out that way; a batch
 * still queued by drprg_hip_map_device_async is completed first).  d_covg given: d_prg_reads == d_covg + n_covg takes the same
 * single call, anything else two calls in one group; the CALLER orders the reduce behind the batch that filled the buffers -- except the batch
 * still queued by drprg_hip_map_device_async INTO these buffers, which the reduce completes first (it may still need the host).  `comm` is an ncclComm_t: a host that already has one (its own RCCL binding) may pass it. */
int drprg_hip_comm_unique_id(uint8_t id[128]);
int drprg_hip_comm_init_rank(void** comm, int nranks, const uint8_t id[128], int rank, int device);
int drprg_hip_comm_destroy(void* comm);
int drprg_hip_allreduce(drprg_hip_ctx* ctx, void* comm, void* d_covg, void* d_prg_reads, void* hip_stream);

void drprg_hip_close(drprg_hip_ctx* ctx);
const char* drprg_hip_last_error(const drprg_hip_ctx* ctx);
/* 1: the library was built with `make EXPERIMENTAL=1` -- it also holds the kernel forms that are bit-exact but measured slower than the
 * defaults (the wave form of read_cluster, refine_kernel, in-kernel clustering of sketch_wave_kernel, read-by-read verification), each
 * behind its environment switch (DRPRG_RC_FORM=wave, DRPRG_FILTER_FORM=refine, DRPRG_WAVE_FUSE=1, DRPRG_VERIFY_FORM=read).  0: the
 * default build; those switches do nothing.  Nothing of the reference corresponds to it (test and measurement infrastructure). */
int drprg_hip_experimental(void);

int drprg_hip_set_opts(drprg_hip_ctx* ctx, const drprg_hip_map_opts* opts);
/* The same with the caller's sizeof(drprg_hip_map_opts): -EINVAL when it differs from this library's (header / library mismatch). */
int drprg_hip_set_opts_sized(drprg_hip_ctx* ctx, const drprg_hip_map_opts* opts, size_t opts_size);

/* Read mapping: the loop inside `pandora map` / `pandora discover` that
 * Pandora::genotype_with / discover_with wait on (/root/reference/src/lib.rs:580-642, :513-578).
 * Coverage accumulates in the context until drprg_hip_reset. */
int drprg_hip_map_fastx(drprg_hip_ctx* ctx, const char* reads_path); /* fasta/fastq, plain or .gz */
/* Parser threads used by drprg_hip_map_fastx (the -t that drprg forwards to pandora, /root/reference/src/predict.rs:236-245);
 * default 4.  The file is cut at record boundaries and parsed in parallel into pinned blocks. */
int drprg_hip_set_threads(drprg_hip_ctx* ctx, int threads);
/* Host-only self-check of that ingest: parses the file with `threads` parser threads and returns
 * out[0..4] = reads, bases, order-independent digest (sum of the FNV-1a hashes of the reads), batches, and how gzip input
 * was inflated (0 plain text, 1 BGZF members in parallel, 2 one member in one libdeflate call, 3 zlib streaming, 4 one plain
 * gzip stream inflated by all threads: chunks entered at block boundaries found in the compressed data, csrc/pgunzip.h). */
int drprg_hip_parse_fastx(const char* reads_path, int threads, uint64_t out[5], char* err, size_t err_len);
/* Host-only self-check of way 4 on any gzip file (not only FASTQ): inflates gz_path with `threads` threads and chunks of
 * chunk_bytes compressed bytes (0 = automatic) into out_path; out[0..2] = bytes written, chunks accepted as their threads
 * inflated them, chunks inflated again from the known position.  Member CRC-32s and lengths are checked as gzip does. */
int drprg_hip_gunzip_file(const char* gz_path, int threads, uint64_t chunk_bytes, const char* out_path, uint64_t out[3], char* err, size_t err_len);
int drprg_hip_map_host(drprg_hip_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads);
/* Batch already resident in HBM.  d_bases: ASCII bases of all reads back to back, 16-byte aligned;
 * d_offsets: u64[n_reads+1], d_offsets[0] == 0, d_offsets[n_reads] == n_bases.  d_covg (u32[2*n_knodes]) and
 * d_prg_reads (u32[n_prgs]) may be NULL to use the context's accumulators; hip_stream may be NULL.
 * Stream rule of every call that takes device buffers (map_device*, map_device_packed*, pack_device, allreduce): the work is queued on
 * hip_stream, or -- NULL -- on the context's own stream, which does NOT wait for the legacy default stream or anybody else's.  Whatever
 * produced the buffers must be complete on that stream: a producer on another stream (a framework's, say) is synchronised by the caller first. */
int drprg_hip_map_device(drprg_hip_ctx* ctx, const void* d_bases, const void* d_offsets, uint64_t n_reads,
    uint64_t n_bases, void* d_covg, void* d_prg_reads, void* hip_stream);
/* The same without the host waiting for the batch: the launch sequence is queued and the call returns; the read-back the
 * sequence ends with (overflow flags, reads left to the generic pipeline, counters) is looked at while the NEXT batch runs
 * -- by the next call, by drprg_hip_sync, or by anything that reads results -- so back-to-back batches leave no gap on the
 * device.  The buffers of a batch (bases, offsets, accumulators) stay valid and unchanged until the call after the next
 * one returns or drprg_hip_sync does; from then on the batch is done on the device as well (its accumulators may be read or
 * reused from any stream).  An error of a batch is reported by the call that completes it. */
int drprg_hip_map_device_async(drprg_hip_ctx* ctx, const void* d_bases, const void* d_offsets, uint64_t n_reads,
    uint64_t n_bases, void* d_covg, void* d_prg_reads, void* hip_stream);
int drprg_hip_sync(drprg_hip_ctx* ctx); /* completes the batch in flight and waits for its stream */

/* ---- 2-bit packed reads: a second input format (SURVEY.md section 8f NEXT-4 "optional 2-bit packing"; the reference takes any fasta /
 * fastq, /root/reference/src/predict.rs:166-170 -- this is about what crosses PCIe and what the streaming kernel reads, a quarter of the
 * bytes).  Base i of a batch is bits [2 (i & 15) + 1 : 2 (i & 15)] of u32 word i >> 4; the letter is bits 2:1 of the base's ASCII
 * code (A 0, C 1, T 2, G 3, either case -- and of any other byte too); npos lists, ascending, the positions of the bases that are
 * not one of ACGTacgt (they invalidate every k-mer that holds them, as in the ASCII form).  Offsets stay u64[n_reads + 1] in BASES.
 * Results are identical to the ASCII form's, bit for bit (the parity suite runs every case through both).
 *   drprg_hip_set_input_format(ctx, 1): drprg_hip_map_fastx packs on its parser threads (0: ASCII blocks, the default).
 *   drprg_hip_pack_reads: host helper, bases[n_bases] -> words[ceil(n_bases / 16)] + npos (-EOVERFLOW, *n_npos set, when npos_cap is
 *     too small).
 *   drprg_hip_map_host_packed / drprg_hip_map_device_packed(_async): the packed counterparts of map_host / map_device(_async); d_words
 *     16-byte aligned like d_bases (the kernels read them with 16-byte loads; -EINVAL otherwise), d_npos may be NULL when n_npos == 0.
 *   drprg_hip_pack_device: the same conversion for a batch that is already on the device (harnesses); d_npos: room for npos_cap
 *     positions, *n_npos receives their number (positions ascending); d_npos NULL: -EINVAL if the batch holds a base that is not ACGT. */
int drprg_hip_set_input_format(drprg_hip_ctx* ctx, int packed);
int drprg_hip_pack_reads(const uint8_t* bases, uint64_t n_bases, uint32_t* words, uint64_t* npos, uint64_t npos_cap, uint64_t* n_npos);
int drprg_hip_map_host_packed(drprg_hip_ctx* ctx, const uint32_t* words, const uint64_t* offsets, uint64_t n_reads, const uint64_t* npos, uint64_t n_npos);
int drprg_hip_map_device_packed(drprg_hip_ctx* ctx, const void* d_words, const void* d_offsets, uint64_t n_reads, uint64_t n_bases, const void* d_npos,
    uint64_t n_npos, void* d_covg, void* d_prg_reads, void* hip_stream);
int drprg_hip_map_device_packed_async(drprg_hip_ctx* ctx, const void* d_words, const void* d_offsets, uint64_t n_reads, uint64_t n_bases, const void* d_npos,
    uint64_t n_npos, void* d_covg, void* d_prg_reads, void* hip_stream);
int drprg_hip_pack_device(drprg_hip_ctx* ctx, const void* d_bases, uint64_t n_bases, void* d_words, void* d_npos, uint64_t npos_cap, uint64_t* n_npos,
    void* hip_stream);

/* The per-k-mer-node coverage vector (what gets sum-reduced across GPUs):
 * covg[2g] forward, covg[2g+1] reverse coverage of global k-mer node g; prg_reads[p] clusters on PRG p. */
int drprg_hip_coverage_size(const drprg_hip_ctx* ctx, uint64_t* n_covg, uint64_t* n_prgs);
int drprg_hip_coverage(drprg_hip_ctx* ctx, uint32_t* covg, uint64_t n_covg, uint32_t* prg_reads, uint64_t n_prgs);
int drprg_hip_set_coverage(drprg_hip_ctx* ctx, const uint32_t* covg, uint64_t n_covg, const uint32_t* prg_reads,
    uint64_t n_prgs, uint64_t total_bases);
int drprg_hip_device_coverage(drprg_hip_ctx* ctx, void** d_covg, void** d_prg_reads);
int drprg_hip_reset(drprg_hip_ctx* ctx);
/* out[0..7] = reads, bases, minimizers examined (direct kernel: every read minimizer; filtered kernel: those that are
 * index keys), hits, clusters kept, hits kept, sequence in use (1 direct + generic cluster pipeline, 2 Bloom-prefiltered,
 * 3 direct in its candidate form), reads of sequences 2 / 3 that went through the generic cluster pipeline instead of the
 * per-read kernel */
int drprg_hip_counters(drprg_hip_ctx* ctx, uint64_t out[8]);

/* Coverage -> VCF: the tail of `pandora map --genotype --local --vcf-refs <genes.fa>`; writes the file
 * Pandora::vcf_filename names (/root/reference/src/lib.rs:644-646), format of
 * /root/reference/tests/cases/predict/ERR4796933.pandora.vcf. */
int drprg_hip_genotype(drprg_hip_ctx* ctx, const char* vcf_refs, const char* out_vcf, const char* sample);
/* exp_depth_covg / min_kmer_covg / #present / #records of the last drprg_hip_genotype */
int drprg_hip_genotype_info(const drprg_hip_ctx* ctx, uint32_t out[4]);

/* The tail of `pandora discover` (Pandora::discover_with, /root/reference/src/lib.rs:513-578) on the coverage accumulated so
 * far: calls every site, walks the called consensus of every present locus and writes <out_dir>/candidate_regions.tsv (the
 * low-coverage regions pandora would hand to its local assembler), <out_dir>/denovo_paths.txt and denovo_sequences.fa.
 * This entry does not look at the reads again: denovo_paths.txt reports "0 loci with denovo variants" (the format
 * /root/reference/src/lib.rs:648-697 parses); drprg_hip_discover_reads below also finds the novel variants.
 * *n_candidates (may be NULL) receives the number of candidate regions. */
int drprg_hip_discover(drprg_hip_ctx* ctx, const char* vcf_refs, const char* out_dir, const char* sample, uint32_t* n_candidates);
/* drprg_hip_discover + the second half of `pandora discover`: a host-side pass over the reads file piles up, per candidate
 * region, what the reads spell between the exact 15-base anchors either side of it; the most frequent allele that differs
 * from the called consensus (>= 3 reads, >= half of the spanning reads) is a novel variant (with illumina = 0 in
 * the map options the strings are aligned to the consensus and counted column by column: >= 4 reads, >= 60 %).  Writes candidate_regions.tsv, denovo_variants.tsv,
 * denovo_sequences.fa and denovo_paths.txt; list_loci != 0 lists the loci and their variants in denovo_paths.txt in pandora's
 * layout (/root/reference/src/lib.rs:3010-3038) so that the caller's make_prg update runs, 0 keeps "0 loci with denovo
 * variants".  out[0..2] = candidate regions, novel variants, loci with novel variants. */
int drprg_hip_discover_reads(drprg_hip_ctx* ctx, const char* reads_path, const char* vcf_refs, const char* out_dir, const char* sample,
    int list_loci, uint32_t out[3]);
/* Reads that stay in HBM.  The reference reads the sample once per pandora process it spawns -- `pandora discover`
 * (/root/reference/src/predict.rs:248-256), then `pandora map` on the updated PRG (:296-302) -- and pandora discover itself walks
 * the reads twice (mapping, then local assembly).  A sample is 1-5 GB of bases and the device has 288 GB:
 *   drprg_hip_keep_reads(ctx, max_bytes): from now on drprg_hip_map_fastx leaves every block it copies to the device there, up to
 *     max_bytes per device (0 = off, the default; past the limit everything kept is dropped and later calls read files again).
 *   drprg_hip_discover_reads then finds the reads that hold an anchor k-mer with one kernel over the resident base stream
 *     (anchor_scan.hip) and runs its pile-up on those alone, when every read mapped since the last reset came from ONE
 *     drprg_hip_map_fastx call on the same path; otherwise it reads the file, as before.  Same output either way.
 *   drprg_hip_map_resident(ctx, from): maps the reads `from` keeps against the index of `ctx` (same devices, in the same order;
 *     `from` stays open during the call).  -ENODATA (-61) when `from` does not hold all of its reads: map the file instead.
 *   drprg_hip_resident_info: out[0] = 1 if every read mapped since the last reset is resident, out[1] = bytes kept (all devices),
 *     out[2] = blocks kept, out[3] = 1 if the last drprg_hip_discover_reads took its reads from HBM. */
int drprg_hip_keep_reads(drprg_hip_ctx* ctx, uint64_t max_bytes);
int drprg_hip_map_resident(drprg_hip_ctx* ctx, drprg_hip_ctx* from);
int drprg_hip_resident_info(drprg_hip_ctx* ctx, uint64_t out[4]);

/* What MakePrg::update does in the reference (/root/reference/src/lib.rs:279-456: mafft --add of the consensus with the novel
 * variants, then make_prg from_msa) for a host without make_prg / mafft: writes the context's PRG file again with every novel variant
 * added as a new site -- first allele: the stretch of the PRG string the variant touches, whole sites it runs into included; second
 * allele: the called path over that stretch with the variant applied; markers numbered again in pandora's parse order -- so that the
 * PRG spells everything it spelled before and the sample's sequence (index it with drprg_hip_index, open it, map again).
 *   drprg_hip_update_prg: the variants of the last drprg_hip_discover_reads.
 *   drprg_hip_update_prg_from_paths: the loci, called paths and variants of a denovo_paths.txt -- this library's or pandora discover's own
 *     (layout: /root/reference/src/lib.rs:3010-3038; the file MakePrg::update is handed, src/predict.rs:260-279).  -EINVAL if the file
 *     does not parse, names a locus the PRG file does not hold, or its node intervals are not intervals of this context's PRG.  Host only.
 * *n_applied: variants placed. */
int drprg_hip_update_prg(drprg_hip_ctx* ctx, const char* out_prg, uint32_t* n_applied);
int drprg_hip_update_prg_from_paths(drprg_hip_ctx* ctx, const char* denovo_paths, const char* out_prg, uint32_t* n_applied);

/* Coverage hand-over between `discover` and the `map` that follows it on the same reads and PRG
 * (/root/reference/src/predict.rs:248-255, :296-302): save writes vector + counters under `tag`; load returns 0 and installs
 * them only if the file exists, is intact and carries the same tag and index shape (-ENOENT otherwise: map the reads). */
int drprg_hip_save_coverage(drprg_hip_ctx* ctx, const char* path, const char* tag);
int drprg_hip_load_coverage(drprg_hip_ctx* ctx, const char* path, const char* tag, uint64_t counters[8]);

/* a-9 on raw inputs, for parity harnesses (host only, no context): the same functions drprg_hip_genotype applies per allele
 * and per site (pandora SampleInfo; pinned by the reference's fixture VCFs, tests/golden/likelihood_kat.tsv).
 * out[0..5] = MEAN_FWD, MEAN_REV, MED_FWD, MED_REV, SUM_FWD, SUM_REV; *gaps = GAPS. */
int drprg_hip_allele_stats(const uint32_t* fwd, const uint32_t* rev, uint32_t n, uint32_t min_kmer_covg, uint32_t out[6], double* gaps);
int drprg_hip_genotype_site(const uint32_t* mean_fwd, const uint32_t* mean_rev, const double* gaps, uint32_t n_alleles, double e,
    double eps, double* likelihood, int32_t* gt, double* gt_conf);
/* Which k-mer nodes every allele of the last drprg_hip_genotype took its statistics over: one TSV line per allele
 * (chrom, pos, allele, n, comma-separated global k-mer node numbers = indexes/2 into the coverage vector). */
int drprg_hip_genotype_alleles(drprg_hip_ctx* ctx, const char* out_tsv);

/* Host-side check of the Bloom filters of the prefiltered kernel (no device needed): out[0] = index k-mer codes tested
 * (both orientations), out[1..3] = codes that level 0 / levels 1+2 / the second stage would wrongly reject (false
 * negatives: must be 0), out[4..6] = bits set per thousand in those three arrays, out[7] = codes that the array holding level 0
 * and the second-stage bits together (second stage inside the streaming kernel) would wrongly reject.  All zero: no filter. */
int drprg_hip_filter_selfcheck(const drprg_hip_ctx* ctx, uint64_t out[8]);
/* ---- between the read loop and the VCF: the coverage model of the sample, the best path of a locus, the presence rule ----
 * (csrc/params.h; what `pandora map` computes in estimate_parameters / find_max_path / add_consensus_path_to_fastaq behind
 * /root/reference/src/lib.rs:580-642.  The reference consumes e = exp_depth_covg through every LIKELIHOOD / GT_CONF --
 * /root/reference/src/filter.rs:12-16, :149 -- and the presence of a locus through the ##contig lines,
 * /root/reference/src/predict.rs:757-765.)  drprg_hip_genotype runs all of it; these entries expose the pieces so that the test
 * suite can hold each against the oracle's separate statement (oracle/oracle_params.c).
 * estimate_parameters: kmer_covg = fwd + rev coverage of every k-mer of every locus with a cluster; out[0] exp_depth_covg,
 * [1] binomial model in force, [2] e_rate, [3] nb_p, [4] nb_r, [5] branch (1 binomial, 2 negative binomial, 3 insufficient
 * coverage, 0 no locus), [6] mean, [7] variance, [8] clusters per locus, [9] binomial p.  coverage_model: the same of the last
 * drprg_hip_genotype, then [10] thresh, [11] loci dropped because their best path was almost bare. */
int drprg_hip_estimate_parameters(const uint32_t* kmer_covg, uint64_t n, uint64_t clusters, uint64_t loci, uint32_t global_covg, int k,
    double e_rate, int bin, double out[10]);
int drprg_hip_kmer_log_prob(int use_bin, double nb_p, double nb_r, double bin_p, uint32_t fwd, uint32_t rev, uint32_t locus_reads, float* out);
int drprg_hip_prob_threshold(const float* logp, uint64_t n, int* out);
/* best path of locus `prg` for per-node log probabilities logp[n_nodes of that locus]: node ids, source and sink excluded */
int drprg_hip_max_path(const drprg_hip_ctx* ctx, uint32_t prg, const float* logp, int thresh, uint32_t max_kmers_to_average, uint32_t* path,
    uint64_t cap, uint64_t* n_path);
/* per-base coverage of the local nodes along `path`; covg2 = (fwd, rev) per k-mer node of the locus */
int drprg_hip_path_base_coverage(const drprg_hip_ctx* ctx, uint32_t prg, const uint32_t* path, uint64_t n_path, const uint32_t* covg2,
    uint32_t* out, uint64_t cap, uint64_t* n_out);
int drprg_hip_path_coverage_too_low(const uint32_t* base_covg, uint64_t n, uint32_t global_covg); /* 1: the locus is dropped */
int drprg_hip_coverage_model(const drprg_hip_ctx* ctx, double out[12]);

/* Index introspection for harnesses: sizes[0..4] = keys, records, prgs, k-mer nodes, table slots. */
int drprg_hip_index_sizes(const drprg_hip_ctx* ctx, uint64_t sizes[5]);
int drprg_hip_index_export(const drprg_hip_ctx* ctx, uint64_t* keys, uint32_t* rec_off, uint32_t* rec_prg,
    uint32_t* rec_knode, uint8_t* rec_strand, uint32_t* prg_min_path_len, uint32_t* prg_knode_base);

/* Device-side tables of an open context: out[0] = 32-bit words of the L2-resident Bloom tier in front of the probe table
 * (0: the table fits the L2, no tier), out[1] = bytes of the open-addressed probe table (keys + slot records), out[2] = bytes
 * of the LDS-resident filter arrays of the prefiltered sequence (0: index too large / k > 15), out[3] = sequence in use,
 * out[4] = bytes of the filter tiers of that sequence that live in global memory (L2-resident: the middle tier's bitmap and
 * code filter; the small tier's block filter, which is the second stage for packed batches; 0 otherwise), out[5] = 0 (reserved). */
int drprg_hip_device_tables(drprg_hip_ctx* ctx, uint64_t out[6]);

/* Local-graph introspection of PRG `prg` (node intervals are offsets into the PRG string, markers and their
 * spaces included: the convention of pandora's denovo_paths.txt, /root/reference/src/lib.rs:3015-3023).
 * Writes up to cap node [start,end) pairs; *n_nodes receives the node count, *n_sites the site count. */
int drprg_hip_prg_nodes(const drprg_hip_ctx* ctx, uint32_t prg, uint32_t* starts, uint32_t* ends, uint32_t cap,
    uint32_t* n_nodes, uint32_t* n_sites);

/* ---- post-VCF stage (host only; SURVEY.md section 8f NEXT-1) ------------------------------------------------
 * Options of Filterer (/root/reference/src/filter.rs:165-197) and MinorAllele (/root/reference/src/minor.rs:19-49)
 * with the CLI defaults of `drprg predict`; a disabled filter is min_covg < 0, max_covg = INT32_MAX,
 * min_strand_bias / min_gt_conf / min_frs < 0, max_indel < 0. */
typedef struct drprg_hip_annotate_opts {
    int32_t min_covg, max_covg;
    float min_strand_bias, min_gt_conf, min_frs;
    int32_t max_indel;
    float maf, max_gaps, max_called_gaps, max_gaps_diff;
    int32_t minor_min_covg;
    float minor_min_strand_bias;
    int32_t ignore_synonymous;
    uint64_t id_seed; /* 0: random 8-hex record IDs like the reference's Uuid::new_v4()[..8] */
} drprg_hip_annotate_opts;

/* Replaces Predict::predict_from_pandora_vcf (/root/reference/src/predict.rs:420-544): pandora VCF -> filtered,
 * annotated VCF (FILTER, PDP/OGT/VARID/PREDICT INFO).  index_dir holds .config.toml, genes.fa, panel.bcf, rules.csv.
 * Deviation: the output is VCF text, not BCF.  err receives the message on failure (may be NULL). */
int drprg_hip_annotate(const char* index_dir, const char* pandora_vcf, const char* out_vcf,
    const drprg_hip_annotate_opts* opts, char* err, size_t err_len);
/* The annotated VCF as BCF2.2 in BGZF: the format and file name (<sample>.drprg.bcf) the reference writes through rust-htslib
 * (/root/reference/src/predict.rs:429-431).  `drprg predict` of this build writes both the text VCF and this. */
int drprg_hip_vcf_to_bcf(const char* vcf_path, const char* bcf_path, char* err, size_t err_len);
/* Replaces Predict::vcf_to_json (/root/reference/src/predict.rs:716-1086).  padding < 0 / index_version NULL: taken from
 * <index_dir>/.config.toml. */
int drprg_hip_report_json(const char* index_dir, const char* annotated_vcf, const char* out_json, const char* sample,
    int padding, const char* index_version, char* err, size_t err_len);

/* HIP-event timing of the dominant kernel (sketch_filter_kernel / sketch_probe_kernel) on its launch stream (bench.py roofline);
 * launches = launches of that kernel (a batch cut into several read ranges counts one per range).
 * enable != 0 starts/keeps timing; ms_total / launches may be NULL; reset != 0 clears the sums. */
int drprg_hip_kernel_timing(drprg_hip_ctx* ctx, int enable, int reset, double* ms_total, uint64_t* launches);
/* How sketch_filter_kernel handed its tiles out in the batch completed last, and when its four wave classes were through (bench.py prints
 * it next to the roofline; csrc/kernels.h FilterSched).  out[0] rounds of the schedule (1: static, one chunk per wave), [1] chunks = slices,
 * [2] tiles per wave of round 0, [3..6] the classes' shares of round 0 in 1/256 of an even one, [7..10] the classes' end times in 10 ns from
 * the kernel's first wave (0: not clocked), [11] chunks per workgroup, [12..19] chunk size of every round in tiles.  Synchronises. */
int drprg_hip_filter_schedule(drprg_hip_ctx* ctx, uint64_t out[20]);

#ifdef __cplusplus
}
#endif
#endif
