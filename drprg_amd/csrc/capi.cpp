#include <set>
#include <atomic>
#include <mutex>
#include <thread>
#include <fstream>
#include <cstdio>
// capi.cpp -- the C ABI declared in include/drprg_hip.h.
#include "../../include/drprg_hip.h"
#include "fastx.h"
#include "genotype.h"
#include "ingest.h"
#include "pgunzip.h"
#include "denovo.h"
#include "mapper.h"
#include "rccl_dyn.h"
#include "pack.h"
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <memory>

using namespace drprg;

// Page-locked ingest blocks, kept between calls AND between contexts of the process: page-locking costs driver time, in series
// over the threads that ask -- a `drprg predict` that finds a novel variant opens a second context on the updated PRG and maps
// the reads again, and paid for pinning the same blocks a second time while they were kept per context.  (Portable allocations: valid for every
// device.)  The blocks go back to the driver when the last context of the process closes.
struct PinPool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> free_, busy_;
    int contexts = 0;
    static PinPool& get()
    {
        static PinPool* p = new PinPool; // (never destroyed: contexts may outlive static destruction order)
        return *p;
    }
    void context_opened()
    {
        std::lock_guard<std::mutex> g(mu);
        ++contexts;
    }
    void context_closed()
    {
        std::vector<std::pair<void*, size_t>> drop;
        {
            std::lock_guard<std::mutex> g(mu);
            if (--contexts > 0) return;
            drop.swap(free_);
        }
        for (auto& b : drop) Mapper::pinned_free(b.first);
    }
    void* take(size_t n)
    {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < free_.size(); ++i)
                if (free_[i].second == n) {
                    void* p = free_[i].first;
                    free_.erase(free_.begin() + (long)i);
                    busy_.emplace_back(p, n);
                    return p;
                }
        }
        void* p = Mapper::pinned_alloc(n);
        if (p) {
            std::lock_guard<std::mutex> g(mu);
            busy_.emplace_back(p, n);
        }
        return p;
    }
    void give_back(void* p)
    {
        std::lock_guard<std::mutex> g(mu);
        for (size_t i = 0; i < busy_.size(); ++i)
            if (busy_[i].first == p) {
                free_.push_back(busy_[i]);
                busy_.erase(busy_.begin() + (long)i);
                return;
            }
        Mapper::pinned_free(p);
    }
};

struct drprg_hip_ctx {
    PrgIndex index;
    std::unique_ptr<Mapper> mapper; // null for a host-only context; device 0 of a multi-device context
    std::vector<std::unique_ptr<Mapper>> extra; // devices 1 .. ndev-1 of drprg_hip_open_multi (map_fastx shards the reads over all)
    MapCounters extra_counts;                   // what the extra devices counted (folded in when their coverage is)
    MapParams params;
    std::string prg_file;
    std::string last_error;
    // host copy of the coverage (set by set_coverage, or downloaded lazily)
    std::vector<uint32_t> covg, prg_reads;
    bool host_coverage_valid = false;
    uint64_t total_bases = 0;
    int threads = 4; // parser threads of drprg_hip_map_fastx
    bool packed_input = false; // drprg_hip_set_input_format: map_fastx packs the reads to 2 bits on the parser threads
    uint32_t ginfo[4] = { 0, 0, 0, 0 };
    std::vector<VcfRecord> last_records; // of the last drprg_hip_genotype (drprg_hip_genotype_alleles)
    CoverageModel last_model;            // of the last drprg_hip_genotype (drprg_hip_coverage_model)
    size_t last_dropped = 0;             // loci it dropped for a bare best path
    // the last drprg_hip_discover_reads: what drprg_hip_update_prg applies
    GenotypeResult last_discover;
    std::vector<NovelVariant> last_variants;
    // drprg_hip_keep_reads: the files drprg_hip_map_fastx has mapped since the last reset (the resident reads stand for a file
    // only if they are that file's and nothing else's); whether the last drprg_hip_discover_reads took its reads from HBM
    std::vector<std::string> mapped_paths;
    bool last_discover_resident = false;
    // (the page-locked ingest blocks of drprg_hip_map_fastx are recycled process-wide: PinPool above)
    // multi-device context: RCCL communicators of its devices (created on first use; empty when RCCL is not used)
    std::vector<Rccl::Comm> comms;
    std::string reduce_how; // how the last drprg_hip_reduce / drprg_hip_allreduce summed the vectors (drprg_hip_reduce_info)
    bool needs_reset = false; // a reduce failed half way: the devices' vectors are in no defined state until drprg_hip_reset
    ~drprg_hip_ctx()
    {
        if (!comms.empty())
            if (const Rccl* r = Rccl::get())
                for (Rccl::Comm c : comms)
                    if (c) (void)r->CommDestroy(c);
        PinPool::get().context_closed();
    }
    drprg_hip_ctx() { PinPool::get().context_opened(); }
};

static thread_local std::string g_last_error;

#define API_BEGIN(ctx)                                  \
    if (!(ctx)) return DRPRG_EINVAL;                    \
    try {
#define API_END(ctx)                                    \
    }                                                   \
    catch (const Error& e) {                            \
        (ctx)->last_error = e.what();                   \
        return e.code;                                  \
    }                                                   \
    catch (const std::bad_alloc&) {                     \
        (ctx)->last_error = "out of host memory";       \
        return DRPRG_ENOMEM;                            \
    }                                                   \
    catch (const std::exception& e) {                   \
        (ctx)->last_error = e.what();                   \
        return DRPRG_EIO;                               \
    }                                                   \
    return DRPRG_OK;

static void apply_defaults(MapParams& p, const drprg_hip_map_opts* o)
{
    p.illumina = o && o->illumina;
    p.error_rate = (o && o->error_rate > 0) ? o->error_rate : (p.illumina ? 0.001 : 0.11);
    p.max_diff = (o && o->max_diff > 0) ? o->max_diff : (p.illumina ? 2 * p.k + 1 : 250);
    p.min_cluster_size = o ? o->min_cluster_size : 10;
    p.genome_size = (o && o->genome_size) ? o->genome_size : 5000000;
    p.genotyping_error_rate = (o && o->genotyping_error_rate > 0) ? o->genotyping_error_rate : 0.01;
    p.kernel_mode = o ? o->kernel : 0;
    p.binomial = o && o->binomial;
}

// Sum of the per-device coverage vectors of a multi-device context into its first device, on the devices (SURVEY.md section 8e:
// "one ncclReduce(sum, u32) after the last batch").  All devices distinct and RCCL present: one communicator over them
// (ncclCommInitAll, kept by the context) and two ncclReduce calls (coverage, reads per PRG) per device inside one group.  A
// device listed twice (what the one-GPU tests do), no RCCL, or DRPRG_HIP_NO_RCCL=1: peer copy into device 0 + an add kernel,
// device after device.  Then the other devices' vectors are cleared and their counters folded into the context.
static void reduce_devices(drprg_hip_ctx* ctx)
{
    if (ctx->extra.empty() || !ctx->mapper) return;
    Mapper& m = *ctx->mapper;
    std::vector<Mapper*> all { &m };
    for (auto& e : ctx->extra) all.push_back(e.get());
    for (Mapper* x : all) x->sync();
    bool distinct = true;
    for (size_t i = 0; i < all.size(); ++i)
        for (size_t j = i + 1; j < all.size(); ++j) distinct &= all[i]->device() != all[j]->device();
    const char* no = std::getenv("DRPRG_HIP_NO_RCCL");
    std::string why;
    const Rccl* r = (distinct && !(no && *no && *no != '0')) ? Rccl::get(&why) : nullptr;
    bool done = false;
    auto chk = [&](int rc, const char* what) {
        if (rc != Rccl::Success) throw Error(DRPRG_EIO, std::string("RCCL ") + what + ": " + (r && r->GetErrorString ? r->GetErrorString(rc) : "error"));
    };
    if (r) {
        if (ctx->comms.empty()) {
            std::vector<int> devs;
            for (Mapper* x : all) devs.push_back(x->device());
            std::vector<Rccl::Comm> comms(all.size(), nullptr);
            const int rc = r->CommInitAll(comms.data(), (int)all.size(), devs.data());
            if (rc == Rccl::Success) ctx->comms = comms;
            else { // no communicator (nothing has been touched yet): the device add path below
                why = std::string("ncclCommInitAll: ") + (r->GetErrorString ? r->GetErrorString(rc) : "error");
                std::fprintf(stderr, "drprg-hip: warning: %s; summing the devices' vectors by peer copies\n", why.c_str());
                r = nullptr;
            }
        }
    }
    if (r) {
        // one ncclReduce per device over [coverage | reads per PRG] (one allocation: Mapper).  A group that was opened is
        // always closed: the first failure is remembered and thrown behind ncclGroupEnd, and the context is marked so that
        // nothing reads vectors of which some may be reduced and others not (drprg_hip_reset clears the mark).
        const size_t nv = 2 * (size_t)m.n_knodes() + m.n_prgs();
        chk(r->GroupStart(), "ncclGroupStart");
        int first_rc = Rccl::Success;
        const char* first_what = "";
        for (size_t d = 0; d < all.size() && first_rc == Rccl::Success; ++d) {
            if (hipSetDevice(all[d]->device()) != hipSuccess) {
                first_rc = -1;
                first_what = "hipSetDevice";
                break;
            }
            first_rc = r->Reduce(all[d]->d_covg(), all[d]->d_covg(), nv, Rccl::Uint32, Rccl::Sum, 0, ctx->comms[d], all[d]->stream());
            first_what = "ncclReduce";
        }
        const int end_rc = r->GroupEnd();
        if (first_rc != Rccl::Success || end_rc != Rccl::Success) {
            ctx->needs_reset = true;
            if (first_rc == -1) throw Error(DRPRG_EIO, "hipSetDevice failed inside the reduce (drprg_hip_reset the context)");
            chk(first_rc, first_what);
            chk(end_rc, "ncclGroupEnd");
        }
        for (Mapper* x : all) {
            if (hipSetDevice(x->device()) != hipSuccess || hipStreamSynchronize(x->stream()) != hipSuccess) throw Error(DRPRG_EIO, "stream synchronisation after ncclReduce failed");
        }
        ctx->reduce_how = "rccl: ncclReduce(sum, u32) over " + std::to_string(all.size()) + " devices";
        done = true;
    }
    if (!done) {
        for (auto& e : ctx->extra) m.add_vectors_from(*e);
        ctx->reduce_how = std::string("device add: peer copy + add kernel per device (") + (distinct ? (why.empty() ? "RCCL switched off" : why) : "a device is listed twice") + ")";
    }
    for (auto& e : ctx->extra) {
        const MapCounters k = e->counters();
        ctx->extra_counts.reads += k.reads; ctx->extra_counts.bases += k.bases; ctx->extra_counts.minimizers += k.minimizers;
        ctx->extra_counts.hits += k.hits; ctx->extra_counts.clusters_kept += k.clusters_kept;
        ctx->extra_counts.hits_kept += k.hits_kept; ctx->extra_counts.leftover_reads += k.leftover_reads;
        e->reset_coverage(false); // (its vectors are in the sum now; the reads it keeps in HBM stay)
    }
    ctx->host_coverage_valid = false;
}

extern "C" {

int drprg_hip_index(const char* prg_file, int w, int k, int threads)
{
    if (!prg_file) return DRPRG_EINVAL;
    try {
        PrgIndex::build_and_save(prg_file, w, k, threads > 0 ? threads : 1);
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return DRPRG_EIO;
    }
    return DRPRG_OK;
}

static drprg_hip_ctx* open_impl(const char* prg_file, int w, int k, int device, bool from_files, int threads, const int* more_devices = nullptr,
    int n_more = 0)
{
    if (!prg_file) {
        g_last_error = "null PRG path";
        return nullptr;
    }
    std::unique_ptr<drprg_hip_ctx> ctx(new (std::nothrow) drprg_hip_ctx);
    if (!ctx) return nullptr;
    // the HIP runtime starts (first call of the process: 50-250 ms) while the index files are read
    std::thread warm;
    if (device >= 0) warm = std::thread(Mapper::warm_device, device);
    struct Join {
        std::thread& t;
        ~Join()
        {
            if (t.joinable()) t.join();
        }
    } join_warm { warm };
    try {
        ctx->prg_file = prg_file;
        if (from_files) {
            // The k-mer graphs and the .idx beside the PRG are this build's own files.  An index directory made by the real
            // pandora holds files of the same names in pandora's private format: they are not parsed, the graphs are rebuilt
            // from the PRG string (tens of milliseconds for an mtb-sized panel), unless DRPRG_HIP_STRICT_INDEX is set.
            try {
                ctx->index.load(prg_file, w, k);
            } catch (const Error& e) {
                const char* strict = std::getenv("DRPRG_HIP_STRICT_INDEX");
                if ((e.code != DRPRG_EFORMAT && e.code != DRPRG_ENOENT) || (strict && *strict && *strict != '0')) throw;
                std::fprintf(stderr, "drprg-hip: warning: %s; rebuilding the k-mer graphs from %s\n", e.what(), prg_file);
                ctx->index = PrgIndex();
                ctx->index.build(prg_file, w, k, threads > 0 ? threads : 1);
            }
        } else ctx->index.build(prg_file, w, k, threads > 0 ? threads : 1);
        ctx->params.w = w;
        ctx->params.k = k;
        apply_defaults(ctx->params, nullptr);
        if (warm.joinable()) warm.join();
        if (device >= 0) ctx->mapper.reset(new Mapper(ctx->index.flat, ctx->params, device));
        for (int i = 0; i < n_more; ++i) ctx->extra.emplace_back(new Mapper(ctx->index.flat, ctx->params, more_devices[i]));
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return nullptr;
    }
    return ctx.release();
}

drprg_hip_ctx* drprg_hip_open(const char* prg_file, int w, int k, int device)
{
    return open_impl(prg_file, w, k, device, true, 1);
}

drprg_hip_ctx* drprg_hip_open_prg(const char* prg_file, int w, int k, int device, int threads)
{
    return open_impl(prg_file, w, k, device, false, threads);
}

drprg_hip_ctx* drprg_hip_open_multi(const char* prg_file, int w, int k, const int* devices, int ndev, int from_files)
{
    if (!devices || ndev < 1) {
        g_last_error = "drprg_hip_open_multi needs at least one device";
        return nullptr;
    }
    for (int i = 0; i < ndev; ++i)
        if (devices[i] < 0) {
            g_last_error = "drprg_hip_open_multi: negative device id";
            return nullptr;
        }
    return open_impl(prg_file, w, k, devices[0], from_files != 0, 4, devices + 1, ndev - 1);
}

void drprg_hip_close(drprg_hip_ctx* ctx) { delete ctx; }

const char* drprg_hip_last_error(const drprg_hip_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

int drprg_hip_set_opts(drprg_hip_ctx* ctx, const drprg_hip_map_opts* opts)
{
    API_BEGIN(ctx)
    apply_defaults(ctx->params, opts);
    if (ctx->mapper) ctx->mapper->set_params(ctx->params);
    for (auto& m : ctx->extra) m->set_params(ctx->params);
    API_END(ctx)
}

int drprg_hip_set_opts_sized(drprg_hip_ctx* ctx, const drprg_hip_map_opts* opts, size_t opts_size)
{
    static_assert(sizeof(drprg_hip_map_opts) == DRPRG_HIP_MAP_OPTS_SIZE, "drprg_hip_map_opts changed size: bump DRPRG_HIP_MAP_OPTS_SIZE");
    if (!ctx) return DRPRG_EINVAL;
    if (opts && opts_size != sizeof(drprg_hip_map_opts)) {
        ctx->last_error = "drprg_hip_map_opts: the caller's struct has " + std::to_string(opts_size) + " bytes, this library's "
            + std::to_string(sizeof(drprg_hip_map_opts)) + " (header / library mismatch)";
        return DRPRG_EINVAL;
    }
    return drprg_hip_set_opts(ctx, opts);
}

static Mapper& need_mapper(drprg_hip_ctx* ctx)
{
    if (!ctx->mapper) throw Error(DRPRG_ENODEV, "host-only context: the hot path runs on a HIP device only (no CPU fallback)");
    return *ctx->mapper;
}

int drprg_hip_map_fastx(drprg_hip_ctx* ctx, const char* reads_path)
{
    API_BEGIN(ctx)
    if (!reads_path) throw Error(DRPRG_EINVAL, "null reads path");
    Mapper& m = need_mapper(ctx);
    ctx->host_coverage_valid = false;
    ctx->mapped_paths.push_back(reads_path);
    // multi-threaded ingest into pinned blocks (ingest.cpp); multi-line FASTQ falls back to the serial reader
    IngestHooks hooks;
    hooks.packed = ctx->packed_input;
    auto host_batch = [](const PinnedBatch& b) {
        Mapper::HostBatch hb;
        hb.bases = b.bases;
        hb.offsets = b.offsets;
        hb.n_reads = b.n_reads;
        hb.packed = b.packed;
        hb.npos = b.npos;
        hb.n_npos = b.n_npos;
        return hb;
    };
    // page-locked ingest blocks are kept by the process between calls and contexts (PinPool)
    hooks.alloc = [](size_t n) -> void* { return PinPool::get().take(n); };
    hooks.release = [](void* p) { PinPool::get().give_back(p); };
    // One device: one submitter at a time (the ingest serialises the calls).  Several devices (drprg_hip_open_multi): the
    // reads shard by block -- a block goes to the first idle device, round robin from the one after the last choice --
    // and the parser threads that carry the blocks are the submitters, one per device at a time.
    const size_t ndev = 1 + ctx->extra.size();
    std::vector<std::mutex> dev_mu(ndev);
    std::atomic<size_t> next_dev { 0 };
    auto mapper_of = [&](size_t d) -> Mapper& { return d == 0 ? m : *ctx->extra[d - 1]; };
    if (ndev == 1) {
        // (the copy of a block overlaps the kernels of the block before it: Mapper::map_host_async)
        hooks.submit = [&](const PinnedBatch& b) { m.map_host_async(host_batch(b)); };
    } else {
        hooks.concurrent_submit = true;
        hooks.submit = [&](const PinnedBatch& b) {
            const size_t first = next_dev.fetch_add(1) % ndev;
            for (size_t i = 0; i < ndev; ++i) {
                const size_t d = (first + i) % ndev;
                std::unique_lock<std::mutex> l(dev_mu[d], std::try_to_lock);
                if (!l.owns_lock()) continue;
                mapper_of(d).map_host_async(host_batch(b));
                return;
            }
            std::lock_guard<std::mutex> l(dev_mu[first]); // all busy: wait for the round-robin choice
            mapper_of(first).map_host_async(host_batch(b));
        };
    }
    // the coverage vectors of the other devices are summed into device 0 ON THE DEVICE (drprg_hip_reduce: one RCCL reduce over
    // the devices of the context, or a peer copy + add kernel per device); unsigned sums commute, so the result does not depend
    // on which device mapped which block
    auto fold = [&]() {
        m.sync();
        for (auto& e : ctx->extra) e->sync();
        reduce_devices(ctx);
    };
    try {
        IngestStats st = ingest_fastx(reads_path, ctx->threads, hooks);
        ctx->total_bases += st.bases;
        fold();
    } catch (const Error& e) {
        if (e.code != DRPRG_EAGAIN_SERIAL) throw; // (a failed multi-device pass leaves partial vectors on the devices: reset before reuse)
        FastxReader rd(reads_path);
        ReadBatch batch;
        while (rd.next_batch(batch, 8u << 20, 1ull << 30)) {
            m.map_host(batch.bases.data(), batch.offsets.data(), batch.n_reads());
            ctx->total_bases += batch.bases.size();
        }
    }
    API_END(ctx)
}

int drprg_hip_set_threads(drprg_hip_ctx* ctx, int threads)
{
    if (!ctx) return DRPRG_EINVAL;
    ctx->threads = threads > 0 ? threads : 1;
    return DRPRG_OK;
}

int drprg_hip_map_host(drprg_hip_ctx* ctx, const uint8_t* bases, const uint64_t* offsets, uint64_t n_reads)
{
    API_BEGIN(ctx)
    if (n_reads && (!bases || !offsets)) throw Error(DRPRG_EINVAL, "null buffer");
    Mapper& m = need_mapper(ctx);
    if (n_reads) {
        m.map_host(bases, offsets, n_reads);
        ctx->total_bases += offsets[n_reads];
    }
    ctx->host_coverage_valid = false;
    API_END(ctx)
}

// ---- 2-bit packed reads (SURVEY.md section 8f NEXT-4) ----
int drprg_hip_set_input_format(drprg_hip_ctx* ctx, int packed)
{
    if (!ctx) return DRPRG_EINVAL;
    ctx->packed_input = packed != 0;
    return DRPRG_OK;
}

int drprg_hip_pack_reads(const uint8_t* bases, uint64_t n_bases, uint32_t* words, uint64_t* npos, uint64_t npos_cap, uint64_t* n_npos)
{
    if ((n_bases && (!bases || !words)) || !n_npos) return DRPRG_EINVAL;
    try {
        // (pack_append works on 64-bit words and writes one word past the last base: through a buffer of its own)
        std::vector<uint64_t> w64(n_bases / 32 + 2, 0);
        std::vector<uint64_t> np;
        uint64_t n = 0;
        if (n_bases) pack_append(w64.data(), n, reinterpret_cast<const char*>(bases), n_bases, np);
        std::memcpy(words, w64.data(), (n_bases + 15) / 16 * sizeof(uint32_t));
        *n_npos = np.size();
        if (np.size() > npos_cap) return DRPRG_EOVERFLOW;
        if (!np.empty()) {
            if (!npos) return DRPRG_EINVAL;
            std::memcpy(npos, np.data(), np.size() * sizeof(uint64_t));
        }
    } catch (const std::bad_alloc&) {
        return DRPRG_ENOMEM;
    }
    return DRPRG_OK;
}

int drprg_hip_map_host_packed(drprg_hip_ctx* ctx, const uint32_t* words, const uint64_t* offsets, uint64_t n_reads, const uint64_t* npos, uint64_t n_npos)
{
    API_BEGIN(ctx)
    if (n_reads && (!words || !offsets)) throw Error(DRPRG_EINVAL, "null buffer");
    if (n_npos && !npos) throw Error(DRPRG_EINVAL, "n_npos > 0 without the positions");
    Mapper& m = need_mapper(ctx);
    if (n_reads) {
        Mapper::HostBatch hb;
        hb.bases = reinterpret_cast<const uint8_t*>(words);
        hb.offsets = offsets;
        hb.n_reads = n_reads;
        hb.packed = true;
        hb.npos = npos;
        hb.n_npos = n_npos;
        m.map_host(hb);
        ctx->total_bases += offsets[n_reads];
    }
    ctx->host_coverage_valid = false;
    API_END(ctx)
}

static int map_device_packed(drprg_hip_ctx* ctx, const void* d_words, const void* d_offsets, uint64_t n_reads, uint64_t n_bases, const void* d_npos,
    uint64_t n_npos, void* d_covg, void* d_prg_reads, void* hip_stream, bool deferred)
{
    API_BEGIN(ctx)
    Mapper& m = need_mapper(ctx);
    m.map_device_packed((const uint32_t*)d_words, (const uint64_t*)d_offsets, n_reads, n_bases, (const uint64_t*)d_npos, n_npos, (uint32_t*)d_covg,
        (uint32_t*)d_prg_reads, (hipStream_t)hip_stream, deferred);
    ctx->total_bases += n_bases;
    ctx->host_coverage_valid = false;
    API_END(ctx)
}

int drprg_hip_map_device_packed(drprg_hip_ctx* ctx, const void* d_words, const void* d_offsets, uint64_t n_reads, uint64_t n_bases, const void* d_npos,
    uint64_t n_npos, void* d_covg, void* d_prg_reads, void* hip_stream)
{
    return map_device_packed(ctx, d_words, d_offsets, n_reads, n_bases, d_npos, n_npos, d_covg, d_prg_reads, hip_stream, false);
}

int drprg_hip_map_device_packed_async(drprg_hip_ctx* ctx, const void* d_words, const void* d_offsets, uint64_t n_reads, uint64_t n_bases, const void* d_npos,
    uint64_t n_npos, void* d_covg, void* d_prg_reads, void* hip_stream)
{
    return map_device_packed(ctx, d_words, d_offsets, n_reads, n_bases, d_npos, n_npos, d_covg, d_prg_reads, hip_stream, true);
}

int drprg_hip_pack_device(drprg_hip_ctx* ctx, const void* d_bases, uint64_t n_bases, void* d_words, void* d_npos, uint64_t npos_cap, uint64_t* n_npos,
    void* hip_stream)
{
    API_BEGIN(ctx)
    Mapper& m = need_mapper(ctx);
    if (!n_npos) throw Error(DRPRG_EINVAL, "null n_npos");
    *n_npos = m.pack_on_device((const uint8_t*)d_bases, n_bases, (uint32_t*)d_words, (uint64_t*)d_npos, npos_cap, (hipStream_t)hip_stream);
    if (*n_npos > npos_cap) throw Error(DRPRG_EOVERFLOW, "more non-ACGT bases than the position buffer holds");
    API_END(ctx)
}

int drprg_hip_map_device(drprg_hip_ctx* ctx, const void* d_bases, const void* d_offsets, uint64_t n_reads, uint64_t n_bases,
    void* d_covg, void* d_prg_reads, void* hip_stream)
{
    API_BEGIN(ctx)
    Mapper& m = need_mapper(ctx);
    m.map_device((const uint8_t*)d_bases, (const uint64_t*)d_offsets, n_reads, n_bases, (uint32_t*)d_covg,
        (uint32_t*)d_prg_reads, (hipStream_t)hip_stream);
    ctx->total_bases += n_bases;
    ctx->host_coverage_valid = false;
    API_END(ctx)
}

int drprg_hip_map_device_async(drprg_hip_ctx* ctx, const void* d_bases, const void* d_offsets, uint64_t n_reads, uint64_t n_bases,
    void* d_covg, void* d_prg_reads, void* hip_stream)
{
    API_BEGIN(ctx)
    Mapper& m = need_mapper(ctx);
    m.map_device_async((const uint8_t*)d_bases, (const uint64_t*)d_offsets, n_reads, n_bases, (uint32_t*)d_covg,
        (uint32_t*)d_prg_reads, (hipStream_t)hip_stream);
    ctx->total_bases += n_bases;
    ctx->host_coverage_valid = false;
    API_END(ctx)
}

int drprg_hip_sync(drprg_hip_ctx* ctx)
{
    API_BEGIN(ctx)
    if (ctx->mapper) ctx->mapper->sync();
    API_END(ctx)
}

int drprg_hip_reduce(drprg_hip_ctx* ctx)
{
    API_BEGIN(ctx)
    need_mapper(ctx);
    reduce_devices(ctx);
    API_END(ctx)
}

int drprg_hip_reduce_info(const drprg_hip_ctx* ctx, char* out, size_t cap)
{
    if (!ctx || !out || cap == 0) return DRPRG_EINVAL;
    std::snprintf(out, cap, "%s", ctx->reduce_how.c_str());
    return DRPRG_OK;
}

int drprg_hip_comm_unique_id(uint8_t id[128])
{
    if (!id) return DRPRG_EINVAL;
    std::string why;
    const Rccl* r = Rccl::get(&why);
    if (!r) {
        g_last_error = why;
        return DRPRG_ENODEV;
    }
    Rccl::UniqueId u;
    const int rc = r->GetUniqueId(&u);
    if (rc != Rccl::Success) {
        g_last_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(rc);
        return DRPRG_EIO;
    }
    std::memcpy(id, u.internal, sizeof u.internal);
    return DRPRG_OK;
}

int drprg_hip_comm_init_rank(void** comm, int nranks, const uint8_t id[128], int rank, int device)
{
    if (!comm || !id || nranks < 1 || rank < 0 || rank >= nranks || device < 0) return DRPRG_EINVAL;
    std::string why;
    const Rccl* r = Rccl::get(&why);
    if (!r) {
        g_last_error = why;
        return DRPRG_ENODEV;
    }
    if (hipSetDevice(device) != hipSuccess) {
        g_last_error = "hipSetDevice failed";
        return DRPRG_ENODEV;
    }
    Rccl::UniqueId u;
    std::memcpy(u.internal, id, sizeof u.internal);
    Rccl::Comm c = nullptr;
    const int rc = r->CommInitRank(&c, nranks, u, rank);
    if (rc != Rccl::Success) {
        g_last_error = std::string("ncclCommInitRank: ") + r->GetErrorString(rc);
        return DRPRG_EIO;
    }
    *comm = c;
    return DRPRG_OK;
}

int drprg_hip_comm_destroy(void* comm)
{
    if (!comm) return DRPRG_OK;
    const Rccl* r = Rccl::get();
    if (!r) return DRPRG_ENODEV;
    return r->CommDestroy(comm) == Rccl::Success ? DRPRG_OK : DRPRG_EIO;
}

int drprg_hip_experimental(void)
{
#ifdef DRPRG_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}

int drprg_hip_allreduce(drprg_hip_ctx* ctx, void* comm, void* d_covg, void* d_prg_reads, void* hip_stream)
{
    API_BEGIN(ctx)
    Mapper& m = need_mapper(ctx);
    if (!comm) throw Error(DRPRG_EINVAL, "null communicator");
    std::string why;
    const Rccl* r = Rccl::get(&why);
    if (!r) throw Error(DRPRG_ENODEV, why);
    // The context's own accumulators: a batch queued by map_device_async may still need the host (leftover reads) before its
    // vector is final, so it is completed first.  A caller's own buffers: the caller orders the reduce behind the batch that
    // filled them (the contract of drprg_hip_map_device_async says when that batch is done) -- waiting here for the batch in
    // flight would serialise the reduce of batch i with the mapping of batch i+1.  But if the batch in flight is the one that fills
    // the very buffers being reduced, the reduce completes it, as it did before round 4 (its vector may still be waiting for an
    // overflow re-run or the leftover pipeline): the caller cannot have meant to sum an unfinished vector.
    if (!d_covg || m.pending_writes_to((const uint32_t*)d_covg, (const uint32_t*)d_prg_reads)) m.sync();
    if (hipSetDevice(m.device()) != hipSuccess) throw Error(DRPRG_EIO, "hipSetDevice failed");
    uint32_t* c = d_covg ? (uint32_t*)d_covg : m.d_covg();
    uint32_t* p = d_prg_reads ? (uint32_t*)d_prg_reads : m.d_prg_reads();
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : m.stream();
    auto chk = [&](int rc, const char* what) {
        if (rc != Rccl::Success) throw Error(DRPRG_EIO, std::string("RCCL ") + what + ": " + r->GetErrorString(rc));
    };
    const size_t nc = 2 * (size_t)m.n_knodes(), np = m.n_prgs();
    if (p == c + nc) {
        // the sample's additive state is ONE vector [coverage | reads per PRG] -- the context's accumulators are laid out that
        // way, and so is a caller's buffer of drprg_hip_coverage_size's n_covg + n_prgs words -- : a single ncclAllReduce
        chk(r->AllReduce(c, c, nc + np, Rccl::Uint32, Rccl::Sum, comm, st), "ncclAllReduce");
        ctx->reduce_how = "rccl: one ncclAllReduce(sum, u32) of " + std::to_string(nc + np) + " words";
    } else {
        // two separate buffers of the caller's: one group of two (closed whatever happens inside)
        chk(r->GroupStart(), "ncclGroupStart");
        const int rc1 = r->AllReduce(c, c, nc, Rccl::Uint32, Rccl::Sum, comm, st);
        const int rc2 = rc1 == Rccl::Success ? r->AllReduce(p, p, np, Rccl::Uint32, Rccl::Sum, comm, st) : rc1;
        const int rc3 = r->GroupEnd();
        chk(rc1, "ncclAllReduce");
        chk(rc2, "ncclAllReduce");
        chk(rc3, "ncclGroupEnd");
        ctx->reduce_how = "rccl: two grouped ncclAllReduce(sum, u32) (separate buffers)";
    }
    ctx->host_coverage_valid = false; // (asynchronous on the stream: the caller synchronises it, or reads through this context, which does)
    API_END(ctx)
}

int drprg_hip_coverage_size(const drprg_hip_ctx* ctx, uint64_t* n_covg, uint64_t* n_prgs)
{
    if (!ctx) return DRPRG_EINVAL;
    if (n_covg) *n_covg = 2 * (uint64_t)ctx->index.flat.total_knodes();
    if (n_prgs) *n_prgs = ctx->index.prgs.size();
    return DRPRG_OK;
}

static void sync_host_coverage(drprg_hip_ctx* ctx)
{
    if (ctx->needs_reset) throw Error(DRPRG_EIO, "a reduce over the context's devices failed half way: drprg_hip_reset the context and map again");
    if (ctx->host_coverage_valid) return;
    if (ctx->mapper) {
        ctx->mapper->download(ctx->covg, ctx->prg_reads);
    } else {
        ctx->covg.assign(2 * (size_t)ctx->index.flat.total_knodes(), 0);
        ctx->prg_reads.assign(ctx->index.prgs.size(), 0);
    }
    ctx->host_coverage_valid = true;
}

int drprg_hip_coverage(drprg_hip_ctx* ctx, uint32_t* covg, uint64_t n_covg, uint32_t* prg_reads, uint64_t n_prgs)
{
    API_BEGIN(ctx)
    sync_host_coverage(ctx);
    if ((covg && n_covg != ctx->covg.size()) || (prg_reads && n_prgs != ctx->prg_reads.size()))
        throw Error(DRPRG_EINVAL, "coverage buffer size mismatch");
    if (covg) std::memcpy(covg, ctx->covg.data(), ctx->covg.size() * sizeof(uint32_t));
    if (prg_reads) std::memcpy(prg_reads, ctx->prg_reads.data(), ctx->prg_reads.size() * sizeof(uint32_t));
    API_END(ctx)
}

int drprg_hip_set_coverage(drprg_hip_ctx* ctx, const uint32_t* covg, uint64_t n_covg, const uint32_t* prg_reads, uint64_t n_prgs,
    uint64_t total_bases)
{
    API_BEGIN(ctx)
    if (!covg || !prg_reads || n_covg != 2 * (uint64_t)ctx->index.flat.total_knodes() || n_prgs != ctx->index.prgs.size())
        throw Error(DRPRG_EINVAL, "coverage buffer size mismatch");
    ctx->covg.assign(covg, covg + n_covg);
    ctx->prg_reads.assign(prg_reads, prg_reads + n_prgs);
    ctx->host_coverage_valid = true;
    ctx->total_bases = total_bases;
    if (ctx->mapper) ctx->mapper->upload(ctx->covg, ctx->prg_reads);
    API_END(ctx)
}

int drprg_hip_device_coverage(drprg_hip_ctx* ctx, void** d_covg, void** d_prg_reads)
{
    API_BEGIN(ctx)
    Mapper& m = need_mapper(ctx);
    if (d_covg) *d_covg = m.d_covg();
    if (d_prg_reads) *d_prg_reads = m.d_prg_reads();
    API_END(ctx)
}

int drprg_hip_reset(drprg_hip_ctx* ctx)
{
    API_BEGIN(ctx)
    if (ctx->mapper) ctx->mapper->reset_coverage();
    for (auto& m : ctx->extra) m->reset_coverage();
    ctx->extra_counts = MapCounters();
    ctx->covg.clear();
    ctx->prg_reads.clear();
    ctx->host_coverage_valid = false;
    ctx->total_bases = 0;
    ctx->mapped_paths.clear();
    ctx->needs_reset = false;
    API_END(ctx)
}

static std::vector<Mapper*> mappers_of(drprg_hip_ctx* ctx)
{
    std::vector<Mapper*> all;
    if (ctx->mapper) all.push_back(ctx->mapper.get());
    for (auto& e : ctx->extra) all.push_back(e.get());
    return all;
}

// do the reads in HBM stand for the file `path` (null: for whatever was mapped)?
static bool reads_resident(drprg_hip_ctx* ctx, const char* path)
{
    const std::vector<Mapper*> all = mappers_of(ctx);
    if (all.empty()) return false;
    for (Mapper* m : all)
        if (!m->kept_complete()) return false;
    if (path && (ctx->mapped_paths.size() != 1 || ctx->mapped_paths[0] != path)) return false;
    return true;
}

int drprg_hip_keep_reads(drprg_hip_ctx* ctx, uint64_t max_bytes)
{
    API_BEGIN(ctx)
    need_mapper(ctx);
    for (Mapper* m : mappers_of(ctx)) m->keep_reads(max_bytes);
    // (reads mapped before this call are not resident: only a context that has mapped nothing yet can stand for a file)
    if (ctx->total_bases != 0) ctx->mapped_paths.assign(2, std::string());
    API_END(ctx)
}

int drprg_hip_map_resident(drprg_hip_ctx* ctx, drprg_hip_ctx* from)
{
    API_BEGIN(ctx)
    if (!from || from == ctx) throw Error(DRPRG_EINVAL, "drprg_hip_map_resident needs another open context");
    need_mapper(ctx);
    const std::vector<Mapper*> dst = mappers_of(ctx), src = mappers_of(from);
    if (!reads_resident(from, nullptr)) throw Error(DRPRG_ENODATA, "the other context does not hold all of its reads in device memory");
    if (dst.size() != src.size()) throw Error(DRPRG_EINVAL, "the two contexts span different numbers of devices");
    for (size_t i = 0; i < dst.size(); ++i)
        if (dst[i]->device() != src[i]->device()) throw Error(DRPRG_EINVAL, "the two contexts list different devices");
    ctx->host_coverage_valid = false;
    for (size_t i = 0; i < dst.size(); ++i) dst[i]->map_kept_from(*src[i]);
    ctx->total_bases += from->total_bases;
    ctx->mapped_paths.insert(ctx->mapped_paths.end(), from->mapped_paths.begin(), from->mapped_paths.end());
    reduce_devices(ctx);
    API_END(ctx)
}

int drprg_hip_resident_info(drprg_hip_ctx* ctx, uint64_t out[4])
{
    API_BEGIN(ctx)
    if (!out) throw Error(DRPRG_EINVAL, "null output");
    out[0] = reads_resident(ctx, nullptr) ? 1 : 0;
    out[1] = out[2] = 0;
    for (Mapper* m : mappers_of(ctx)) {
        out[1] += m->kept_bytes();
        out[2] += m->kept().size();
    }
    out[3] = ctx->last_discover_resident ? 1 : 0;
    API_END(ctx)
}

int drprg_hip_counters(drprg_hip_ctx* ctx, uint64_t out[8])
{
    API_BEGIN(ctx)
    if (!out) throw Error(DRPRG_EINVAL, "null output");
    std::memset(out, 0, 8 * sizeof(uint64_t));
    if (ctx->mapper) {
        MapCounters c = ctx->mapper->counters();
        const MapCounters& x = ctx->extra_counts; // (the other devices of a multi-device context)
        out[0] = c.reads + x.reads; out[1] = c.bases + x.bases; out[2] = c.minimizers + x.minimizers; out[3] = c.hits + x.hits;
        out[4] = c.clusters_kept + x.clusters_kept; out[5] = c.hits_kept + x.hits_kept; out[6] = c.kernel; out[7] = c.leftover_reads + x.leftover_reads;
    }
    API_END(ctx)
}

int drprg_hip_genotype(drprg_hip_ctx* ctx, const char* vcf_refs, const char* out_vcf, const char* sample)
{
    API_BEGIN(ctx)
    if (!out_vcf) throw Error(DRPRG_EINVAL, "null output path");
    sync_host_coverage(ctx);
    GenotypeResult r = genotype(ctx->index, ctx->covg, ctx->prg_reads, ctx->total_bases, ctx->params, vcf_refs ? vcf_refs : "");
    write_vcf(out_vcf, r, sample && *sample ? sample : "sample");
    ctx->ginfo[0] = r.exp_depth_covg;
    ctx->ginfo[1] = r.min_kmer_covg;
    ctx->ginfo[2] = (uint32_t)r.present.size();
    ctx->ginfo[3] = (uint32_t)r.records.size();
    ctx->last_model = r.model;
    ctx->last_dropped = r.dropped_low_coverage.size();
    ctx->last_records = std::move(r.records);
    API_END(ctx)
}

// ---- discover (SURVEY.md section 8f NEXT-2) ---------------------------------------------------------------
int drprg_hip_discover(drprg_hip_ctx* ctx, const char* vcf_refs, const char* out_dir, const char* sample, uint32_t* n_candidates)
{
    API_BEGIN(ctx)
    if (!out_dir) throw Error(DRPRG_EINVAL, "null output directory");
    sync_host_coverage(ctx);
    GenotypeResult r = genotype(ctx->index, ctx->covg, ctx->prg_reads, ctx->total_bases, ctx->params, vcf_refs ? vcf_refs : "");
    const std::string dir = out_dir, smp = sample && *sample ? sample : "sample";
    {
        std::ofstream o(dir + "/candidate_regions.tsv");
        o << "#locus\tstart\tend\tlow_start\tlow_end\tmax_covg\tconsensus\n";
        for (const CandidateRegion& c : r.candidates)
            o << c.chrom << "\t" << c.start << "\t" << c.end << "\t" << c.low_start << "\t" << c.low_end << "\t" << c.max_covg << "\t" << c.seq << "\n";
        if (!o) throw Error(DRPRG_EIO, "cannot write " + dir + "/candidate_regions.tsv");
    }
    {
        // pandora's denovo_paths.txt surface (/root/reference/src/lib.rs:648-697 parses "<N> loci with denovo variants" and the
        // line before every "<n> nodes" line).  Local assembly of the candidate regions is not implemented: no locus is ever
        // reported as carrying a novel variant, so MakePrg::update keeps the index PRG (/root/reference/src/lib.rs:299-301).
        std::ofstream o(dir + "/denovo_paths.txt");
        o << "1 samples\nSample " << smp << "\n0 loci with denovo variants\n";
        if (!o) throw Error(DRPRG_EIO, "cannot write " + dir + "/denovo_paths.txt");
        std::ofstream f(dir + "/denovo_sequences.fa");
    }
    if (n_candidates) *n_candidates = (uint32_t)r.candidates.size();
    API_END(ctx)
}

int drprg_hip_discover_reads(drprg_hip_ctx* ctx, const char* reads_path, const char* vcf_refs, const char* out_dir, const char* sample,
    int list_loci, uint32_t out[3])
{
    API_BEGIN(ctx)
    if (!out_dir || !reads_path) throw Error(DRPRG_EINVAL, "null argument");
    sync_host_coverage(ctx);
    const DiscoverParams dp;
    GenotypeResult r = genotype(ctx->index, ctx->covg, ctx->prg_reads, ctx->total_bases, ctx->params, vcf_refs ? vcf_refs : "", dp);
    const std::string dir = out_dir, smp = sample && *sample ? sample : "sample";
    {
        std::ofstream o(dir + "/candidate_regions.tsv");
        o << "#locus\tstart\tend\tlow_start\tlow_end\tmax_covg\tconsensus\n";
        for (const CandidateRegion& c : r.candidates)
            o << c.chrom << "\t" << c.start << "\t" << c.end << "\t" << c.low_start << "\t" << c.low_end << "\t" << c.max_covg << "\t" << c.seq << "\n";
        if (!o) throw Error(DRPRG_EIO, "cannot write " + dir + "/candidate_regions.tsv");
    }
    // accurate reads (-I): whole strings between the anchors are counted; noisy reads: column-wise majority of their alignments
    DiscoverParams adp = dp;
    if (!ctx->params.illumina) {
        adp.min_support = 4;
        adp.min_fraction = 0.6;
    }
    // the reads of this very file are in HBM (drprg_hip_keep_reads): the device picks the few that hold an anchor k-mer
    ResidentReads resident;
    ctx->last_discover_resident = reads_resident(ctx, reads_path);
    if (ctx->last_discover_resident)
        resident = [ctx](const std::vector<uint64_t>& anchors, uint32_t A, std::vector<uint8_t>& bases, std::vector<uint64_t>& offsets) {
            for (Mapper* m : mappers_of(ctx)) m->select_reads_with_anchors(anchors, A, bases, offsets);
        };
    std::vector<NovelVariant> variants = assemble_candidate_regions(r, reads_path, ctx->threads, adp, ctx->params.illumina, resident);
    write_denovo_paths(dir, smp, r, variants, list_loci != 0);
    if (out) {
        out[0] = (uint32_t)r.candidates.size();
        out[1] = (uint32_t)variants.size();
        std::set<std::string> loci;
        for (const NovelVariant& v : variants) loci.insert(v.chrom);
        out[2] = (uint32_t)loci.size();
    }
    ctx->last_discover = std::move(r);
    ctx->last_variants = std::move(variants);
    API_END(ctx)
}

int drprg_hip_update_prg(drprg_hip_ctx* ctx, const char* out_prg, uint32_t* n_applied)
{
    API_BEGIN(ctx)
    if (!out_prg) throw Error(DRPRG_EINVAL, "null output path");
    std::vector<std::pair<std::string, std::string>> prgs;
    for (const LocalGraph& g : ctx->index.prgs) prgs.emplace_back(g.name, g.prg);
    std::vector<std::string> skipped;
    const uint32_t n = update_prgs(prgs, ctx->last_discover, ctx->last_variants, &skipped);
    for (const std::string& s : skipped)
        std::fprintf(stderr, "drprg-hip: novel variant at %s could not be placed in the PRG\n", s.c_str());
    std::ofstream o(out_prg);
    for (auto& p : prgs) o << ">" << p.first << "\n" << p.second << "\n";
    if (!o) throw Error(DRPRG_EIO, std::string("cannot write ") + out_prg);
    if (n_applied) *n_applied = n;
    API_END(ctx)
}

int drprg_hip_update_prg_from_paths(drprg_hip_ctx* ctx, const char* denovo_paths, const char* out_prg, uint32_t* n_applied)
{
    API_BEGIN(ctx)
    if (!denovo_paths || !out_prg) throw Error(DRPRG_EINVAL, "null path");
    std::vector<std::pair<std::string, std::string>> prgs;
    std::vector<std::string> names;
    for (const LocalGraph& g : ctx->index.prgs) {
        prgs.emplace_back(g.name, g.prg);
        names.push_back(g.name);
    }
    GenotypeResult gr;
    std::vector<NovelVariant> variants;
    read_denovo_paths(denovo_paths, names, gr, variants);
    for (const LocusConsensus& lc : gr.consensus) // the node intervals must be this PRG's
        for (const ConsensusNode& n : lc.nodes)
            if (n.end > prgs[lc.prg].second.size() || prgs[lc.prg].second.compare(n.start, n.end - n.start, n.seq) != 0)
                throw Error(DRPRG_EINVAL, std::string(denovo_paths) + ": the nodes of " + lc.chrom + " are not intervals of this context's PRG");
    std::vector<std::string> skipped;
    const uint32_t n = update_prgs(prgs, gr, variants, &skipped);
    for (const std::string& s : skipped) std::fprintf(stderr, "drprg-hip: novel variant at %s could not be placed in the PRG\n", s.c_str());
    std::ofstream o(out_prg);
    for (auto& p : prgs) o << ">" << p.first << "\n" << p.second << "\n";
    if (!o) throw Error(DRPRG_EIO, std::string("cannot write ") + out_prg);
    if (n_applied) *n_applied = n;
    API_END(ctx)
}

// Coverage cache: `pandora discover` and the `pandora map` that follows stream the same reads against the same PRG
// (/root/reference/src/predict.rs:248-255, :296-302); the first saves its vector, the second takes it back if `tag` agrees.
static const char COVG_MAGIC[8] = { 'D', 'R', 'P', 'R', 'G', 'C', 'V', '1' };

int drprg_hip_save_coverage(drprg_hip_ctx* ctx, const char* path, const char* tag)
{
    API_BEGIN(ctx)
    if (!path || !tag) throw Error(DRPRG_EINVAL, "null argument");
    sync_host_coverage(ctx);
    std::ofstream o(path, std::ios::binary);
    const uint64_t hdr[4] = { std::strlen(tag), ctx->covg.size(), ctx->prg_reads.size(), ctx->total_bases };
    uint64_t cnt[8] = { 0 };
    if (ctx->mapper) {
        const MapCounters c = ctx->mapper->counters();
        const uint64_t v[8] = { c.reads, c.bases, c.minimizers, c.hits, c.clusters_kept, c.hits_kept, c.kernel, c.leftover_reads };
        std::memcpy(cnt, v, sizeof cnt);
    }
    o.write(COVG_MAGIC, 8);
    o.write((const char*)hdr, sizeof hdr);
    o.write((const char*)cnt, sizeof cnt);
    o.write(tag, (std::streamsize)hdr[0]);
    o.write((const char*)ctx->covg.data(), (std::streamsize)(ctx->covg.size() * sizeof(uint32_t)));
    o.write((const char*)ctx->prg_reads.data(), (std::streamsize)(ctx->prg_reads.size() * sizeof(uint32_t)));
    if (!o) throw Error(DRPRG_EIO, std::string("cannot write ") + path);
    API_END(ctx)
}

int drprg_hip_load_coverage(drprg_hip_ctx* ctx, const char* path, const char* tag, uint64_t counters[8])
{
    API_BEGIN(ctx)
    if (!path || !tag) throw Error(DRPRG_EINVAL, "null argument");
    std::ifstream in(path, std::ios::binary);
    if (!in) return DRPRG_ENOENT;
    char magic[8];
    uint64_t hdr[4], cnt[8];
    in.read(magic, 8);
    in.read((char*)hdr, sizeof hdr);
    in.read((char*)cnt, sizeof cnt);
    if (!in || std::memcmp(magic, COVG_MAGIC, 8) != 0 || hdr[0] > (1u << 20)) return DRPRG_EFORMAT;
    std::string t(hdr[0], '\0');
    in.read(&t[0], (std::streamsize)hdr[0]);
    if (!in || t != tag || hdr[1] != 2 * (uint64_t)ctx->index.flat.total_knodes() || hdr[2] != ctx->index.prgs.size())
        return DRPRG_ENOENT; // another PRG / other reads / other parameters: not this run's vector
    std::vector<uint32_t> covg(hdr[1]), prg_reads(hdr[2]);
    in.read((char*)covg.data(), (std::streamsize)(covg.size() * sizeof(uint32_t)));
    in.read((char*)prg_reads.data(), (std::streamsize)(prg_reads.size() * sizeof(uint32_t)));
    if (!in) return DRPRG_EFORMAT;
    ctx->covg.swap(covg);
    ctx->prg_reads.swap(prg_reads);
    ctx->host_coverage_valid = true;
    ctx->total_bases = hdr[3];
    if (ctx->mapper) ctx->mapper->upload(ctx->covg, ctx->prg_reads);
    if (counters) std::memcpy(counters, cnt, sizeof cnt);
    API_END(ctx)
}

int drprg_hip_genotype_alleles(drprg_hip_ctx* ctx, const char* out_tsv)
{
    API_BEGIN(ctx)
    if (!out_tsv) throw Error(DRPRG_EINVAL, "null output path");
    std::ofstream o(out_tsv);
    if (!o) throw Error(DRPRG_EIO, std::string("cannot write ") + out_tsv);
    o << "#chrom\tpos\tallele\tn_kmers\tglobal_kmer_nodes\n";
    for (const VcfRecord& rec : ctx->last_records)
        for (size_t a = 0; a < rec.allele_knodes.size(); ++a) {
            o << rec.chrom << "\t" << rec.pos << "\t" << a << "\t" << rec.allele_knodes[a].size() << "\t";
            for (size_t i = 0; i < rec.allele_knodes[a].size(); ++i) o << (i ? "," : "") << rec.allele_knodes[a][i];
            o << "\n";
        }
    if (!o) throw Error(DRPRG_EIO, std::string("short write to ") + out_tsv);
    API_END(ctx)
}

int drprg_hip_estimate_parameters(const uint32_t* kmer_covg, uint64_t n, uint64_t clusters, uint64_t loci, uint32_t global_covg, int k, double e_rate,
    int bin, double out[10])
{
    if ((n && !kmer_covg) || !out) return DRPRG_EINVAL;
    const CoverageModel m = estimate_parameters(std::vector<uint32_t>(kmer_covg, kmer_covg + n), clusters, loci, global_covg, k, e_rate, bin != 0);
    out[0] = m.exp_depth_covg; out[1] = m.bin ? 1 : 0; out[2] = m.e_rate; out[3] = m.nb_p; out[4] = m.nb_r; out[5] = m.branch;
    out[6] = m.mean; out[7] = m.var; out[8] = m.num_reads; out[9] = m.bin_p;
    return DRPRG_OK;
}

int drprg_hip_kmer_log_prob(int use_bin, double nb_p, double nb_r, double bin_p, uint32_t fwd, uint32_t rev, uint32_t locus_reads, float* out)
{
    if (!out) return DRPRG_EINVAL;
    CoverageModel m;
    m.bin = use_bin != 0;
    m.nb_p = (float)nb_p;
    m.nb_r = (float)nb_r;
    m.bin_p = bin_p;
    *out = kmer_log_prob(m, fwd, rev, locus_reads);
    return DRPRG_OK;
}

int drprg_hip_prob_threshold(const float* logp, uint64_t n, int* out)
{
    if ((n && !logp) || !out) return DRPRG_EINVAL;
    *out = prob_threshold(std::vector<float>(logp, logp + n));
    return DRPRG_OK;
}

int drprg_hip_max_path(const drprg_hip_ctx* ctx, uint32_t prg, const float* logp, int thresh, uint32_t max_kmers_to_average, uint32_t* path,
    uint64_t cap, uint64_t* n_path)
{
    if (!ctx || !logp || !n_path || prg >= ctx->index.kgs.size()) return DRPRG_EINVAL;
    const KmerGraph& kg = ctx->index.kgs[prg];
    const std::vector<uint32_t> p = find_max_path(kg, std::vector<float>(logp, logp + kg.nodes.size()), thresh, max_kmers_to_average);
    *n_path = p.size();
    for (size_t i = 0; i < p.size() && i < cap; ++i) path[i] = p[i];
    return DRPRG_OK;
}

int drprg_hip_path_base_coverage(const drprg_hip_ctx* ctx, uint32_t prg, const uint32_t* path, uint64_t n_path, const uint32_t* covg2, uint32_t* out,
    uint64_t cap, uint64_t* n_out)
{
    if (!ctx || (n_path && !path) || !covg2 || !n_out || prg >= ctx->index.kgs.size()) return DRPRG_EINVAL;
    const KmerGraph& kg = ctx->index.kgs[prg];
    for (uint64_t i = 0; i < n_path; ++i)
        if (path[i] >= kg.nodes.size()) return DRPRG_EINVAL;
    const std::vector<uint32_t> b = base_coverage_along_path(ctx->index.prgs[prg], kg, std::vector<uint32_t>(path, path + n_path), covg2);
    *n_out = b.size();
    for (size_t i = 0; i < b.size() && i < cap; ++i) out[i] = b[i];
    return DRPRG_OK;
}

int drprg_hip_path_coverage_too_low(const uint32_t* base_covg, uint64_t n, uint32_t global_covg)
{
    if (n && !base_covg) return DRPRG_EINVAL;
    return path_coverage_too_low(std::vector<uint32_t>(base_covg, base_covg + n), global_covg) ? 1 : 0;
}

int drprg_hip_coverage_model(const drprg_hip_ctx* ctx, double out[12])
{
    if (!ctx || !out) return DRPRG_EINVAL;
    const CoverageModel& m = ctx->last_model;
    out[0] = m.exp_depth_covg; out[1] = m.bin ? 1 : 0; out[2] = m.e_rate; out[3] = m.nb_p; out[4] = m.nb_r; out[5] = m.branch;
    out[6] = m.mean; out[7] = m.var; out[8] = m.num_reads; out[9] = m.bin_p; out[10] = m.thresh; out[11] = (double)ctx->last_dropped;
    return DRPRG_OK;
}

int drprg_hip_allele_stats(const uint32_t* fwd, const uint32_t* rev, uint32_t n, uint32_t min_kmer_covg, uint32_t out[6], double* gaps)
{
    if ((n && (!fwd || !rev)) || !out || !gaps) return DRPRG_EINVAL;
    const AlleleStats s = allele_stats(std::vector<uint32_t>(fwd, fwd + n), std::vector<uint32_t>(rev, rev + n), min_kmer_covg);
    out[0] = s.mean_fwd; out[1] = s.mean_rev; out[2] = s.med_fwd; out[3] = s.med_rev; out[4] = s.sum_fwd; out[5] = s.sum_rev;
    *gaps = s.gaps;
    return DRPRG_OK;
}

int drprg_hip_genotype_site(const uint32_t* mean_fwd, const uint32_t* mean_rev, const double* gaps, uint32_t n_alleles, double e,
    double eps, double* likelihood, int32_t* gt, double* gt_conf)
{
    if (!n_alleles || !mean_fwd || !mean_rev || !gaps || !likelihood || !gt || !gt_conf) return DRPRG_EINVAL;
    std::vector<AlleleStats> al(n_alleles);
    for (uint32_t a = 0; a < n_alleles; ++a) {
        al[a].mean_fwd = mean_fwd[a];
        al[a].mean_rev = mean_rev[a];
        al[a].gaps = gaps[a];
    }
    int g = 0;
    genotype_site(al, e, eps, g, *gt_conf);
    *gt = g;
    for (uint32_t a = 0; a < n_alleles; ++a) likelihood[a] = al[a].likelihood;
    return DRPRG_OK;
}

int drprg_hip_genotype_info(const drprg_hip_ctx* ctx, uint32_t out[4])
{
    if (!ctx || !out) return DRPRG_EINVAL;
    std::memcpy(out, ctx->ginfo, sizeof ctx->ginfo);
    return DRPRG_OK;
}

int drprg_hip_filter_selfcheck(const drprg_hip_ctx* ctx, uint64_t out[8])
{
    if (!ctx || !out) return DRPRG_EINVAL;
    ctx->index.filter_selfcheck(out);
    return DRPRG_OK;
}

int drprg_hip_index_sizes(const drprg_hip_ctx* ctx, uint64_t sizes[5])
{
    if (!ctx || !sizes) return DRPRG_EINVAL;
    const FlatIndex& f = ctx->index.flat;
    sizes[0] = f.keys.size();
    sizes[1] = f.rec_prg.size();
    sizes[2] = ctx->index.prgs.size();
    sizes[3] = f.total_knodes();
    sizes[4] = f.slot_key.size();
    return DRPRG_OK;
}

int drprg_hip_index_export(const drprg_hip_ctx* ctx, uint64_t* keys, uint32_t* rec_off, uint32_t* rec_prg, uint32_t* rec_knode,
    uint8_t* rec_strand, uint32_t* prg_min_path_len, uint32_t* prg_knode_base)
{
    if (!ctx) return DRPRG_EINVAL;
    const FlatIndex& f = ctx->index.flat;
    if (keys) std::memcpy(keys, f.keys.data(), f.keys.size() * sizeof(uint64_t));
    if (rec_off) std::memcpy(rec_off, f.rec_off.data(), f.rec_off.size() * sizeof(uint32_t));
    if (rec_prg) std::memcpy(rec_prg, f.rec_prg.data(), f.rec_prg.size() * sizeof(uint32_t));
    if (rec_knode) std::memcpy(rec_knode, f.rec_knode_global.data(), f.rec_knode_global.size() * sizeof(uint32_t));
    if (rec_strand) std::memcpy(rec_strand, f.rec_strand.data(), f.rec_strand.size());
    if (prg_min_path_len) std::memcpy(prg_min_path_len, f.min_path_len.data(), f.min_path_len.size() * sizeof(uint32_t));
    if (prg_knode_base) std::memcpy(prg_knode_base, f.knode_base.data(), f.knode_base.size() * sizeof(uint32_t));
    return DRPRG_OK;
}

int drprg_hip_prg_nodes(const drprg_hip_ctx* ctx, uint32_t prg, uint32_t* starts, uint32_t* ends, uint32_t cap, uint32_t* n_nodes,
    uint32_t* n_sites)
{
    if (!ctx || prg >= ctx->index.prgs.size()) return DRPRG_EINVAL;
    const LocalGraph& g = ctx->index.prgs[prg];
    for (uint32_t i = 0; i < g.nodes.size() && i < cap; ++i) {
        if (starts) starts[i] = g.nodes[i].start;
        if (ends) ends[i] = g.nodes[i].end;
    }
    if (n_nodes) *n_nodes = (uint32_t)g.nodes.size();
    if (n_sites) *n_sites = (uint32_t)g.sites.size();
    return DRPRG_OK;
}

int drprg_hip_device_tables(drprg_hip_ctx* ctx, uint64_t out[6])
{
    API_BEGIN(ctx)
    need_mapper(ctx).device_tables(out);
    API_END(ctx)
}

int drprg_hip_kernel_timing(drprg_hip_ctx* ctx, int enable, int reset, double* ms_total, uint64_t* launches)
{
    API_BEGIN(ctx)
    Mapper& m = need_mapper(ctx);
    m.sync(); // (a batch in flight carries the events of the setting it was launched with)
    m.enable_kernel_timing(enable != 0);
    if (ms_total) *ms_total = m.sketch_ms_total();
    if (launches) *launches = m.sketch_launches();
    if (reset) m.reset_kernel_timing();
    API_END(ctx)
}

int drprg_hip_filter_schedule(drprg_hip_ctx* ctx, uint64_t out[20])
{
    API_BEGIN(ctx)
    if (!out) throw Error(DRPRG_EINVAL, "null pointer");
    need_mapper(ctx).filter_schedule(out);
    API_END(ctx)
}

} // extern "C"

// ---- post-VCF stage ------------------------------------------------------------------------------------
#include "report.h"

static int report_guard(char* err, size_t err_len, const std::function<void()>& fn)
{
    auto put = [&](const char* m) {
        if (err && err_len) {
            std::strncpy(err, m, err_len - 1);
            err[err_len - 1] = 0;
        }
    };
    try {
        fn();
    } catch (const Error& e) {
        put(e.what());
        return e.code;
    } catch (const std::exception& e) {
        put(e.what());
        return DRPRG_EIO;
    }
    return DRPRG_OK;
}

extern "C" {

int drprg_hip_annotate(const char* index_dir, const char* pandora_vcf, const char* out_vcf, const drprg_hip_annotate_opts* o,
    char* err, size_t err_len)
{
    if (!index_dir || !pandora_vcf || !out_vcf || !o) return DRPRG_EINVAL;
    return report_guard(err, err_len, [&]() {
        report::AnnotateOpts a;
        a.filter.min_covg = o->min_covg;
        a.filter.max_covg = o->max_covg;
        a.filter.min_strand_bias = o->min_strand_bias;
        a.filter.min_gt_conf = o->min_gt_conf;
        a.filter.min_frs = o->min_frs;
        a.filter.has_max_indel = o->max_indel >= 0;
        a.filter.max_indel = o->max_indel;
        a.minor.maf = o->maf;
        a.minor.max_gaps = o->max_gaps;
        a.minor.max_called_gaps = o->max_called_gaps;
        a.minor.max_gaps_diff = o->max_gaps_diff;
        a.minor.minor_min_covg = o->minor_min_covg;
        a.minor.minor_min_strand_bias = o->minor_min_strand_bias;
        a.ignore_synonymous = o->ignore_synonymous != 0;
        a.id_seed = o->id_seed;
        report::annotate_vcf(report::IndexFiles { index_dir }, pandora_vcf, out_vcf, a);
    });
}

int drprg_hip_report_json(const char* index_dir, const char* annotated_vcf, const char* out_json, const char* sample, int padding,
    const char* index_version, char* err, size_t err_len)
{
    if (!index_dir || !annotated_vcf || !out_json) return DRPRG_EINVAL;
    return report_guard(err, err_len, [&]() {
        report::IndexFiles idx { index_dir };
        std::string version = index_version ? index_version : "";
        if (padding < 0 || !index_version) {
            report::IndexConfig c = report::read_config(idx.config());
            if (padding < 0) padding = c.padding;
            if (!index_version) version = c.version;
        }
        report::vcf_to_json(idx, annotated_vcf, out_json, sample ? sample : "sample", padding, version);
    });
}

} // extern "C"

// ---- ingest self-check (host only) ---------------------------------------------------------------------------
extern "C" int drprg_hip_vcf_to_bcf(const char* vcf_path, const char* bcf_path, char* err, size_t err_len)
{
    if (!vcf_path || !bcf_path) return DRPRG_EINVAL;
    return report_guard(err, err_len, [&]() { report::vcf_to_bcf(vcf_path, bcf_path); });
}

extern "C" int drprg_hip_gunzip_file(const char* gz_path, int threads, uint64_t chunk_bytes, const char* out_path, uint64_t out[3], char* err, size_t err_len)
{
    if (!gz_path || !out_path || !out) return DRPRG_EINVAL;
    return report_guard(err, err_len, [&]() {
        const int fd = open(gz_path, O_RDONLY);
        if (fd < 0) throw Error(DRPRG_ENOENT, std::string("cannot open ") + gz_path);
        struct stat sb;
        void* map = fstat(fd, &sb) == 0 && sb.st_size > 0 ? mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
        close(fd);
        if (map == MAP_FAILED) throw Error(DRPRG_EIO, std::string("cannot map ") + gz_path);
        struct Unmap {
            void* p;
            size_t n;
            ~Unmap() { munmap(p, n); }
        } unmap { map, (size_t)sb.st_size };
        FILE* o = std::fopen(out_path, "wb");
        if (!o) throw Error(DRPRG_EIO, std::string("cannot write ") + out_path);
        struct Close {
            FILE* f;
            ~Close() { std::fclose(f); }
        } close_o { o };
        ParallelGunzip pg((const unsigned char*)map, (size_t)sb.st_size, threads, (size_t)chunk_bytes);
        std::vector<char> buf(size_t(8) << 20);
        uint64_t total = 0;
        for (size_t n; (n = pg.read(buf.data(), buf.size())) > 0; total += n)
            if (std::fwrite(buf.data(), 1, n, o) != n) throw Error(DRPRG_EIO, std::string("cannot write ") + out_path);
        out[0] = total;
        out[1] = pg.chunks_accepted();
        out[2] = pg.chunks_redone();
    });
}

extern "C" int drprg_hip_parse_fastx(const char* reads_path, int threads, uint64_t out[5], char* err, size_t err_len)
{
    if (!reads_path || !out) return DRPRG_EINVAL;
    return report_guard(err, err_len, [&]() {
        uint64_t sum = 0, n_reads = 0, n_bases = 0, batches = 0;
        const bool no_digest = std::getenv("DRPRG_PARSE_NO_DIGEST") != nullptr; // (ingest timing without the checksum)
        auto digest = [&](const uint8_t* bases, const uint64_t* offsets, uint64_t n) {
            for (uint64_t i = 0; i < n && !no_digest; ++i) { // order-independent: sum of FNV-1a hashes of the reads
                uint64_t h = 1469598103934665603ull;
                for (uint64_t j = offsets[i]; j < offsets[i + 1]; ++j) h = (h ^ bases[j]) * 1099511628211ull;
                sum += h;
            }
            n_reads += n;
            n_bases += offsets[n];
            ++batches;
        };
        // DRPRG_PARSE_FORMAT (self-check of the packing ingest, no device needed): "normalized" = the digest of the ASCII blocks with every
        // base upper-cased and every byte that is not ACGTacgt read as N; "packed" = the parser threads pack (IngestHooks::packed) and the
        // digest is taken over what the packed block says (letters back to A C G T, N at the positions in npos): the two must be equal
        const char* fmt = std::getenv("DRPRG_PARSE_FORMAT");
        const bool packed = fmt && std::string(fmt) == "packed", normalized = fmt && std::string(fmt) == "normalized";
        std::vector<uint8_t> tmp;
        auto digest_any = [&](const PinnedBatch& b) {
            if (!packed && !normalized) {
                digest(b.bases, b.offsets, b.n_reads);
                return;
            }
            tmp.resize(b.n_bases);
            if (b.packed) {
                const uint32_t* w = reinterpret_cast<const uint32_t*>(b.bases);
                for (uint64_t i = 0; i < b.n_bases; ++i) tmp[i] = (uint8_t)"ACTG"[(w[i >> 4] >> (2 * (i & 15))) & 3u];
                for (uint64_t i = 0; i < b.n_npos; ++i) {
                    if (b.npos[i] >= b.n_bases || (i && b.npos[i] <= b.npos[i - 1])) throw Error(DRPRG_EIO, "packed block: positions not ascending");
                    tmp[b.npos[i]] = 'N';
                }
            } else {
                for (uint64_t i = 0; i < b.n_bases; ++i) {
                    const uint8_t u = b.bases[i] & 0xDFu;
                    tmp[i] = (u == 'A' || u == 'C' || u == 'G' || u == 'T') ? u : (uint8_t)'N';
                }
            }
            digest(tmp.data(), b.offsets, b.n_reads);
        };
        IngestHooks hooks;
        hooks.packed = packed;
        hooks.submit = digest_any;
        out[4] = 0;
        try {
            out[4] = (uint64_t)ingest_fastx(reads_path, threads, hooks).gz_mode;
        } catch (const Error& e) {
            if (e.code != DRPRG_EAGAIN_SERIAL) throw;
            FastxReader rd(reads_path);
            ReadBatch batch;
            while (rd.next_batch(batch, 1u << 20, 1ull << 28)) digest(batch.bases.data(), batch.offsets.data(), batch.n_reads());
        }
        out[0] = n_reads; out[1] = n_bases; out[2] = sum; out[3] = batches;
    });
}
