// index.cpp -- build / save / load the minimizer index.  See index.h.
#include "index.h"
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <sys/stat.h>
#include <thread>

namespace drprg {

static std::string dir_of(const std::string& p)
{
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

std::string PrgIndex::idx_path(const std::string& prg_file, int w, int k)
{
    return prg_file + ".k" + std::to_string(k) + ".w" + std::to_string(w) + ".idx";
}

std::string PrgIndex::gfa_path(const std::string& prg_file, const std::string& name, int w, int k)
{
    return dir_of(prg_file) + "/kmer_prgs/" + name + ".k" + std::to_string(k) + ".w" + std::to_string(w) + ".gfa";
}

void PrgIndex::build(const std::string& prg_file, int w_, int k_, int threads)
{
    if (k_ < 1 || k_ > 32 || w_ < 1) throw Error(DRPRG_EINVAL, "need 1 <= k <= 32 and w >= 1");
    w = w_;
    k = k_;
    prgs = load_prg_file(prg_file);
    kgs.assign(prgs.size(), KmerGraph());
    std::atomic<size_t> next { 0 };
    std::string err;
    std::atomic<bool> failed { false };
    auto work = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < prgs.size();) {
            try {
                kgs[i].build(prgs[i], w, k);
            } catch (const std::exception& e) {
                if (!failed.exchange(true)) err = e.what();
            }
        }
    };
    int nt = std::max(1, std::min<int>(threads, (int)prgs.size()));
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (failed) throw Error(DRPRG_EFORMAT, err);
    flatten();
}

void PrgIndex::flatten()
{
    FlatIndex& f = flat;
    f = FlatIndex();
    f.knode_base.assign(prgs.size() + 1, 0);
    f.min_path_len.assign(prgs.size(), 0);
    struct Rec {
        uint64_t key;
        uint32_t prg, knode;
        uint8_t strand;
    };
    std::vector<Rec> recs;
    for (size_t p = 0; p < prgs.size(); ++p) {
        f.knode_base[p + 1] = f.knode_base[p] + (uint32_t)kgs[p].nodes.size();
        f.min_path_len[p] = kgs[p].shortest_path_length;
        const auto& nodes = kgs[p].nodes;
        for (size_t i = 1; i + 1 < nodes.size(); ++i)
            recs.push_back(Rec { nodes[i].hash, (uint32_t)p, nodes[i].id, (uint8_t)(nodes[i].strand ? 1 : 0) });
    }
    std::sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) {
        if (a.key != b.key) return a.key < b.key;
        if (a.prg != b.prg) return a.prg < b.prg;
        return a.knode < b.knode;
    });
    for (size_t i = 0; i < recs.size(); ++i) {
        if (i == 0 || recs[i].key != recs[i - 1].key) {
            f.keys.push_back(recs[i].key);
            f.rec_off.push_back((uint32_t)i);
        }
        f.rec_prg.push_back(recs[i].prg);
        f.rec_knode_global.push_back(f.knode_base[recs[i].prg] + recs[i].knode);
        f.rec_strand.push_back(recs[i].strand);
    }
    f.rec_off.push_back((uint32_t)recs.size());
    // open-addressed table at <= 50 % load
    uint32_t bits = 4;
    while ((1ULL << bits) < 2 * f.keys.size() + 1) ++bits;
    f.table_bits = bits;
    const size_t nslot = (size_t)1 << bits;
    f.slot_key.assign(nslot, 0);
    f.slot_off.assign(nslot, 0);
    f.slot_cnt.assign(nslot, 0);
    for (size_t i = 0; i < f.keys.size(); ++i) {
        uint32_t s = table_slot(f.keys[i], bits, k <= 15);
        while (f.slot_cnt[s] != 0) s = (s + 1) & (uint32_t)(nslot - 1);
        f.slot_key[s] = f.keys[i];
        f.slot_off[s] = f.rec_off[i];
        f.slot_cnt[s] = f.rec_off[i + 1] - f.rec_off[i];
    }
    // Bloom filter of index k-mer codes for the LDS-prefiltered sketch kernel
    constexpr uint32_t MAX_WBITS = 14; // 16384 words = 64 KB of LDS
    const size_t entries = 2 * recs.size();
    // every index k-mer as the kernels' code (letter = bits 2:1 of the ASCII base: A 0, C 1, T 2, G 3; complement = letter ^ 2;
    // first base in the lowest bits), in both orientations
    auto for_each_code = [&](auto&& fn) {
        for (size_t p = 0; p < prgs.size(); ++p) {
            const auto& nodes = kgs[p].nodes;
            for (size_t i = 1; i + 1 < nodes.size(); ++i) {
                const std::string s = kpath_sequence(prgs[p], nodes[i].path);
                uint32_t fw = 0, rc = 0;
                for (int j = 0; j < k; ++j) {
                    const uint32_t c = ((uint32_t)(unsigned char)s[(size_t)j] >> 1) & 3u;
                    fw |= c << (2 * j);
                    rc |= (c ^ 2u) << (2 * (k - 1 - j));
                }
                fn(fw);
                fn(rc);
            }
        }
    };
    constexpr uint32_t L0_WBITS = 15;
    const bool small_tier = k <= 15 && entries > 0 && entries <= 3 * (size_t(1) << MAX_WBITS);
    const bool level0 = small_tier && k == 15 && 4 * entries * 3 <= (size_t(32) << L0_WBITS) / 2;
    const bool force_mid = k == 15 && std::getenv("DRPRG_FORCE_MID_TIER") != nullptr; // (tests: small panels through the middle tier)
    if (small_tier && !force_mid && (level0 || k < 15 || std::getenv("DRPRG_NO_MID_TIER"))) {
        // level 0 (k = 15 and a small index only): a second array of 2^15 words keyed on the 12-mers at offsets 0..3 of
        // every index k-mer, so that the kernel probes one 12-mer per four read positions (the 12-mer at 4g+3 lies
        // inside every 15-mer that starts at 4g..4g+3).  Levels 1+2 then get 32 KB: 160 KB of LDS in all.
        const uint32_t max_wbits = level0 ? MAX_WBITS - 1 : MAX_WBITS;
        uint32_t wbits = 8;
        while (wbits < max_wbits && (size_t(1) << wbits) * 5 < entries * 4) ++wbits; // <= 1.25 entries per word
        f.bloom_wbits = wbits;
        f.bloom.assign(size_t(1) << wbits, 0);
        // layout: see sketch_filter.hip
        const uint32_t wmask = (1u << wbits) - 1;
        const uint32_t kmask = (1u << (2 * k)) - 1; // k <= 15
        if (level0) {
            f.bloom0_wbits = L0_WBITS;
            f.bloom0.assign(size_t(1) << L0_WBITS, 0);
            f.bloom0f.assign(size_t(1) << L0_WBITS, 0);
            f.bloomr.assign(size_t(1) << BLOOMR_WBITS, 0);
        }
        const uint32_t wmask0 = (1u << L0_WBITS) - 1;
        for_each_code([&](uint32_t code) {
            const uint32_t x = code & kmask & 0xFFFFFFu; // level 1: the first min(k,12) bases, 24 x 24 bit multiply
            const uint32_t h = (uint32_t)((uint64_t)x * BLOOM_C1);
            f.bloom[(h >> 18) & wmask] |= (1u << (31 - (h & 31))) | (1u << (31 - ((h >> 8) & 31))) | (1u << (31 - ((x >> 16) & 31)));
            const uint32_t h2 = code * BLOOM_C2; // level 2: an independent word, tested only for level-1 survivors
            f.bloom[h2 >> (32 - wbits)] |= (1u << (h2 & 31)) | (1u << ((h2 >> 5) & 31)) | (1u << ((h2 >> 10) & 31));
            if (level0) { // second stage: one word, four bits (fill < 15 %: one in a few thousand false positives)
                const uint32_t hr = code * BLOOM_CR;
                f.bloomr[hr >> (32 - BLOOMR_WBITS)] |= (1u << (hr & 31)) | (1u << ((hr >> 5) & 31)) | (1u << ((hr >> 10) & 31)) | (1u << ((hr >> 15) & 31));
                // the same stage inside the level-0 array: word = top 15 bits of the hash, three bits from its low 15, three from a
                // second hash (six bits: the array is a third full, and every false positive is a candidate verify_count_kernel pays for)
                const uint32_t hs = code * BLOOM_C2;
                f.bloom0f[hr >> (32 - L0_WBITS)] |= (1u << (hr & 31)) | (1u << ((hr >> 5) & 31)) | (1u << ((hr >> 10) & 31)) | (1u << (hs >> 27))
                    | (1u << ((hs >> 22) & 31)) | (1u << ((hs >> 17) & 31));
            }
            if (level0)
                for (int o = 0; o < 4; ++o) {
                    const uint32_t y = (code >> (2 * o)) & 0xFFFFFFu;
                    const uint32_t g = (uint32_t)((uint64_t)y * BLOOM_C0);
                    const uint32_t bits0 = (1u << (31 - (g & 31))) | (1u << (31 - ((g >> 8) & 31))) | (1u << (31 - ((y >> 16) & 31)));
                    f.bloom0[(g >> 17) & wmask0] |= bits0;
                    f.bloom0f[(g >> 17) & wmask0] |= bits0;
                }
        });
        if (level0) { // the second stage for the L2 (index.h blkc)
            uint32_t cw = 10;
            while (cw < MID_C_MAX_WBITS && (size_t(4) << cw) < 4 * entries) ++cw;
            f.blkc_wbits = cw;
            f.blkc.assign(size_t(4) << cw, 0);
            for_each_code([&](uint32_t code) {
                const uint32_t h = code * BLOOM_CR;
                for (int o = 0; o < 4; ++o) {
                    const uint32_t key = (code >> (2 * o)) & 0xFFFFFFu;
                    const uint32_t blk = (uint32_t)((uint64_t)key * BLOOM_C0) >> (32 - cw);
                    uint32_t* b = &f.blkc[(size_t)blk * 4];
                    b[0] |= 1u << (h >> 27);
                    b[1] |= 1u << ((h >> 22) & 31);
                    b[2] |= 1u << ((h >> 17) & 31);
                    b[3] |= 1u << ((h >> 12) & 31);
                }
            });
        }
        if (level0 && std::getenv("DRPRG_FT_STATS")) { // how full the level-0 array is, alone and with the second stage's bits in it, and what
            // a random 12-mer's three-bit test passes at in each (mean over the words of (bits / 32)^3: the third bit's position comes from the key itself)
            auto stat = [](const std::vector<uint32_t>& a, const char* name) {
                double bits = 0, p3 = 0;
                for (uint32_t wd : a) {
                    const double b = (double)__builtin_popcount(wd);
                    bits += b;
                    p3 += (b / 32) * (b / 32) * (b / 32);
                }
                std::fprintf(stderr, "[filter] %s: %.1f %% of %zu bits set, a random key passes the three-bit test at %.2f %%\n", name, 100 * bits / (32.0 * a.size()),
                    32 * a.size(), 100 * p3 / a.size());
            };
            stat(f.bloom0, "level 0 alone");
            stat(f.bloom0f, "level 0 + second-stage bits");
        }
    } else if (k == 15 && entries > 0 && !std::getenv("DRPRG_NO_MID_TIER") && recs.size() <= mid_tier_max_records()) {
        // Middle tier (round 3): too many index k-mers for an LDS-resident filter of the whole codes.  Level 0 stays in LDS but
        // is keyed on the CANONICAL 12-mer (half the entries); what passes it is looked up in the exact bitmap of the canonical
        // index 12-mers (2 MB, in the L2: ~267 G four-byte probes per second chip-wide, measured -- tools/mb_l2probe.hip), and
        // the four 15-mer codes of a group that passes both in a one-word Bloom filter of the codes (>= 32 bits per code, four
        // set: one false candidate in several thousand tests).  The forms above are all-LDS and faster: they serve the indexes
        // that fit them.
        f.mid_bitmap.assign(MID_BITMAP_WORDS, 0);
        for_each_code([&](uint32_t code) {
            for (int o = 0; o < 4; ++o) {
                const uint32_t key = canon12_code(code >> (2 * o));
                f.mid_bitmap[key >> 5] |= 1u << (key & 31);
            }
        });
        uint64_t distinct = 0;
        for (uint32_t wd : f.mid_bitmap) distinct += (uint64_t)__builtin_popcount(wd);
        // level 0: three bits per 12-mer while the array stays under ~70 % full, else one (a fuller array rejects less with three)
        f.mid0_bits = distinct * 5 <= (size_t(32) << L0_WBITS) * 2 ? 3 : 1;
        f.mid0.assign(size_t(1) << L0_WBITS, 0);
        const uint32_t wmask0 = (1u << L0_WBITS) - 1;
        for (uint32_t wi = 0; wi < MID_BITMAP_WORDS; ++wi)
            for (uint32_t wd = f.mid_bitmap[wi]; wd; wd &= wd - 1) {
                const uint32_t key = (wi << 5) | (uint32_t)__builtin_ctz(wd);
                const uint32_t g = (uint32_t)((uint64_t)key * BLOOM_C0);
                uint32_t bits0 = 1u << (31 - (g & 31));
                if (f.mid0_bits == 3) bits0 |= (1u << (31 - ((g >> 8) & 31))) | (1u << (31 - ((key >> 16) & 31)));
                f.mid0[(g >> 17) & wmask0] |= bits0;
            }
        // split-block filter of the codes: <= ~8 entries per 16-byte block while the table stays within 2 MB (entries = codes x 4 alignments)
        uint32_t cw = 10;
        while (cw < MID_C_MAX_WBITS && (size_t(8) << cw) < 4 * entries) ++cw;
        f.midc_wbits = cw;
        f.midc.assign(size_t(4) << cw, 0);
        for_each_code([&](uint32_t code) {
            const uint32_t h = code * BLOOM_CR;
            for (int o = 0; o < 4; ++o) {
                const uint32_t key = canon12_code(code >> (2 * o));
                const uint32_t blk = (uint32_t)((uint64_t)key * BLOOM_C0) >> (32 - cw);
                uint32_t* b = &f.midc[(size_t)blk * 4];
                b[0] |= 1u << (h >> 27);
                b[1] |= 1u << ((h >> 22) & 31);
                b[2] |= 1u << ((h >> 17) & 31);
                b[3] |= 1u << ((h >> 12) & 31);
            }
        });
    }
}

void PrgIndex::filter_selfcheck(uint64_t out[8]) const
{
    for (int i = 0; i < 8; ++i) out[i] = 0;
    const FlatIndex& f = flat;
    if (f.midc_wbits) { // middle tier: out[1] / out[2] / out[3] = codes that level 0 / the 12-mer bitmap / the code filter would reject
        const uint32_t wmask0 = (uint32_t)f.mid0.size() - 1;
        for (size_t p = 0; p < prgs.size(); ++p) {
            const auto& nodes = kgs[p].nodes;
            for (size_t i = 1; i + 1 < nodes.size(); ++i) {
                const std::string s = kpath_sequence(prgs[p], nodes[i].path);
                uint32_t both[2] = { 0, 0 };
                for (int j = 0; j < k; ++j) {
                    const uint32_t c = ((uint32_t)(unsigned char)s[(size_t)j] >> 1) & 3u;
                    both[0] |= c << (2 * j);
                    both[1] |= (c ^ 2u) << (2 * (k - 1 - j));
                }
                for (uint32_t code : both) {
                    ++out[0];
                    bool miss0 = false, missb = false;
                    for (int o = 0; o < 4; ++o) { // the kernel sees this k-mer at offset o of a group: key = the 12-mer at 3 - o... any of the four
                        const uint32_t key = canon12_code(code >> (2 * o));
                        const uint32_t g = (uint32_t)((uint64_t)key * BLOOM_C0), word = f.mid0[(g >> 17) & wmask0];
                        uint32_t t = word << (g & 31);
                        if (f.mid0_bits == 3) t &= (word << ((g >> 8) & 31)) & (word << ((key >> 16) & 31));
                        miss0 |= !(t >> 31);
                        missb |= !((f.mid_bitmap[key >> 5] >> (key & 31)) & 1u);
                    }
                    out[1] += miss0;
                    out[2] += missb;
                    const uint32_t h = code * BLOOM_CR;
                    bool missc = false;
                    for (int o = 0; o < 4; ++o) { // whatever the alignment, the block of that alignment's 12-mer must hold the code's bits
                        const uint32_t key = canon12_code(code >> (2 * o));
                        const uint32_t* b = &f.midc[(size_t)((uint32_t)((uint64_t)key * BLOOM_C0) >> (32 - f.midc_wbits)) * 4];
                        missc |= !((b[0] >> (h >> 27)) & (b[1] >> ((h >> 22) & 31)) & (b[2] >> ((h >> 17) & 31)) & (b[3] >> ((h >> 12) & 31)) & 1u);
                    }
                    out[3] += missc;
                }
            }
        }
        auto fill = [](const std::vector<uint32_t>& v) -> uint64_t {
            uint64_t bits = 0;
            for (uint32_t w : v) bits += (uint64_t)__builtin_popcount(w);
            return v.empty() ? 0 : bits * 1000 / (32 * (uint64_t)v.size());
        };
        out[4] = fill(f.mid0);
        out[5] = fill(f.mid_bitmap);
        out[6] = fill(f.midc);
        return;
    }
    if (!f.bloom_wbits) return;
    const uint32_t wmask = (1u << f.bloom_wbits) - 1, kmask = (1u << (2 * k)) - 1, wmask0 = f.bloom0_wbits ? (1u << f.bloom0_wbits) - 1 : 0;
    // the tests exactly as sketch_filter.hip states them (bloom_test: bits 31 - shift of the word)
    auto three = [](uint32_t word, uint32_t h, uint32_t x) {
        return ((word << (h & 31)) & (word << ((h >> 8) & 31)) & (word << ((x >> 16) & 31))) >> 31;
    };
    auto check = [&](uint32_t code) {
        ++out[0];
        if (f.bloom0_wbits) { // the kernel probes the 12-mer at offset 3 of a group of four positions: a 15-mer at position
            bool any_miss = false; // 4g+o inside the group holds that 12-mer at its own offset 3-o
            for (int o = 0; o < 4; ++o) {
                const uint32_t y = (code >> (2 * o)) & 0xFFFFFFu;
                const uint32_t g = (uint32_t)((uint64_t)y * BLOOM_C0);
                any_miss |= !three(f.bloom0[(g >> 17) & wmask0], g, y);
            }
            out[1] += any_miss;
            const uint32_t hr = code * BLOOM_CR, word = f.bloomr[hr >> (32 - BLOOMR_WBITS)];
            out[3] += !((word >> (hr & 31)) & (word >> ((hr >> 5) & 31)) & (word >> ((hr >> 10) & 31)) & (word >> ((hr >> 15) & 31)) & 1u);
            // the shared array of the fused form: both tests
            bool miss_f = false;
            for (int o = 0; o < 4; ++o) {
                const uint32_t y = (code >> (2 * o)) & 0xFFFFFFu;
                const uint32_t g = (uint32_t)((uint64_t)y * BLOOM_C0);
                miss_f |= !three(f.bloom0f[(g >> 17) & wmask0], g, y);
            }
            const uint32_t wf = f.bloom0f[hr >> (32 - f.bloom0_wbits)];
            const uint32_t hs = code * BLOOM_C2;
            miss_f |= !((wf >> (hr & 31)) & (wf >> ((hr >> 5) & 31)) & (wf >> ((hr >> 10) & 31)) & (wf >> (hs >> 27)) & (wf >> ((hs >> 22) & 31))
                & (wf >> ((hs >> 17) & 31)) & 1u);
            // ... and the second stage of the L2 form: every block the code was entered in (the plain 12-mer at offset o picks it) holds its four bits
            bool miss_blk = false;
            if (f.blkc_wbits)
                for (int o = 0; o < 4; ++o) {
                    const uint32_t key = (code >> (2 * o)) & 0xFFFFFFu;
                    const uint32_t* b = &f.blkc[(size_t)((uint32_t)((uint64_t)key * BLOOM_C0) >> (32 - f.blkc_wbits)) * 4];
                    miss_blk |= !((b[0] >> (hr >> 27)) & (b[1] >> ((hr >> 22) & 31)) & (b[2] >> ((hr >> 17) & 31)) & (b[3] >> ((hr >> 12) & 31)) & 1u);
                }
            out[7] += miss_f || miss_blk;
        }
        const uint32_t x = code & kmask & 0xFFFFFFu;
        const uint32_t h = (uint32_t)((uint64_t)x * BLOOM_C1);
        const uint32_t h2 = code * BLOOM_C2, w2 = f.bloom[h2 >> (32 - f.bloom_wbits)];
        const bool l1 = three(f.bloom[(h >> 18) & wmask], h, x);
        const bool l2 = (w2 >> (h2 & 31)) & (w2 >> ((h2 >> 5) & 31)) & (w2 >> ((h2 >> 10) & 31)) & 1u;
        out[2] += !(l1 && l2);
    };
    for (size_t p = 0; p < prgs.size(); ++p) {
        const auto& nodes = kgs[p].nodes;
        for (size_t i = 1; i + 1 < nodes.size(); ++i) {
            const std::string s = kpath_sequence(prgs[p], nodes[i].path);
            uint32_t fw = 0, rc = 0;
            for (int j = 0; j < k; ++j) {
                const uint32_t c = ((uint32_t)(unsigned char)s[(size_t)j] >> 1) & 3u;
                fw |= c << (2 * j);
                rc |= (c ^ 2u) << (2 * (k - 1 - j));
            }
            check(fw);
            check(rc);
        }
    }
    auto fill = [](const std::vector<uint32_t>& v) -> uint64_t {
        uint64_t bits = 0;
        for (uint32_t w : v) bits += (uint64_t)__builtin_popcount(w);
        return v.empty() ? 0 : bits * 1000 / (32 * (uint64_t)v.size());
    };
    out[4] = fill(f.bloom0);
    out[5] = fill(f.bloom);
    out[6] = fill(f.bloomr);
}

void PrgIndex::save(const std::string& prg_file) const
{
    std::string kdir = dir_of(prg_file) + "/kmer_prgs";
    if (mkdir(kdir.c_str(), 0777) != 0 && errno != EEXIST) throw Error(DRPRG_EIO, "cannot create " + kdir);
    for (size_t p = 0; p < prgs.size(); ++p) kgs[p].save_gfa(gfa_path(prg_file, prgs[p].name, w, k), prgs[p]);
    std::ofstream out(idx_path(prg_file, w, k));
    if (!out) throw Error(DRPRG_EIO, "cannot write " + idx_path(prg_file, w, k));
    const FlatIndex& f = flat;
    // pandora's .idx layout as SURVEY.md appendix A.3 records it [UPSTREAM-MEMORY]: first line = number of distinct minimizer
    // keys, then per key: hash <TAB> n <TAB> n records "(prg_id, path, knode_id, strand)", the path printed like the S lines
    // of the k-mer graphs, e.g. 2{[10, 20)[25, 30)}
    auto path_string = [&](uint32_t prg, uint32_t node) {
        std::ostringstream os;
        const KPath& kp = kgs[prg].nodes[node].path;
        os << kp.size() << "{";
        for (const PathPiece& pp : kp) {
            const uint32_t base = prgs[prg].nodes[pp.node].start;
            os << "[" << base + pp.off_start << ", " << base + pp.off_end << ")";
        }
        os << "}";
        return os.str();
    };
    out << f.keys.size() << "\n";
    for (size_t i = 0; i < f.keys.size(); ++i) {
        out << f.keys[i] << "\t" << (f.rec_off[i + 1] - f.rec_off[i]);
        for (uint32_t j = f.rec_off[i]; j < f.rec_off[i + 1]; ++j) {
            const uint32_t prg = f.rec_prg[j], node = f.rec_knode_global[j] - f.knode_base[prg];
            out << "\t(" << prg << ", " << path_string(prg, node) << ", " << node << ", " << (int)f.rec_strand[j] << ")";
        }
        out << "\n";
    }
    if (!out) throw Error(DRPRG_EIO, "short write to " + idx_path(prg_file, w, k));
}

void PrgIndex::build_and_save(const std::string& prg_file, int w, int k, int threads)
{
    PrgIndex idx;
    idx.build(prg_file, w, k, threads);
    idx.save(prg_file);
}

void PrgIndex::load(const std::string& prg_file, int w_, int k_)
{
    w = w_;
    k = k_;
    prgs = load_prg_file(prg_file);
    kgs.assign(prgs.size(), KmerGraph());
    for (size_t p = 0; p < prgs.size(); ++p) kgs[p].load_gfa(gfa_path(prg_file, prgs[p].name, w, k), prgs[p], w, k);
    flatten();
    // the .idx file must agree with what the k-mer graphs imply
    std::ifstream in(idx_path(prg_file, w, k));
    if (!in) throw Error(DRPRG_ENOENT, "cannot open " + idx_path(prg_file, w, k));
    size_t nkeys = 0, nrec = 0;
    in >> nkeys;
    std::string line;
    std::getline(in, line);
    if (nkeys != flat.keys.size())
        throw Error(DRPRG_EFORMAT, idx_path(prg_file, w, k) + " does not match kmer_prgs/ (key count)");
    // every record of the file must be a record the k-mer graphs imply: same key, same PRG, same node, same strand (the
    // path is re-derived from the GFA; a file written by the previous layout of this build -- "prg node strand" -- parses too)
    for (size_t i = 0; i < nkeys; ++i) {
        if (!std::getline(in, line)) throw Error(DRPRG_EFORMAT, idx_path(prg_file, w, k) + " is truncated");
        std::istringstream is(line);
        uint64_t key;
        uint32_t n;
        is >> key >> n;
        if (key != flat.keys[i] || n != flat.rec_off[i + 1] - flat.rec_off[i])
            throw Error(DRPRG_EFORMAT, idx_path(prg_file, w, k) + " does not match kmer_prgs/ (key " + std::to_string(i) + ")");
        std::string field;
        std::getline(is, field, '\t'); // rest of the count column
        for (uint32_t j = flat.rec_off[i]; j < flat.rec_off[i + 1]; ++j) {
            if (!std::getline(is, field, '\t')) throw Error(DRPRG_EFORMAT, idx_path(prg_file, w, k) + ": key " + std::to_string(i) + " lists too few records");
            unsigned prg = 0, node = 0;
            int strand = 0;
            bool ok;
            const size_t brace = field.find('}');
            if (!field.empty() && field[0] == '(' && brace != std::string::npos)
                ok = std::sscanf(field.c_str(), "(%u,", &prg) == 1 && std::sscanf(field.c_str() + brace + 1, ", %u, %d)", &node, &strand) == 2;
            else
                ok = std::sscanf(field.c_str(), "%u %u %d", &prg, &node, &strand) == 3;
            if (!ok || prg != flat.rec_prg[j] || node != flat.rec_knode_global[j] - flat.knode_base[flat.rec_prg[j]] || strand != (int)flat.rec_strand[j])
                throw Error(DRPRG_EFORMAT, idx_path(prg_file, w, k) + " does not match kmer_prgs/ (record " + std::to_string(j) + ")");
        }
        nrec += n;
    }
    (void)nrec;
}

double MapParams::cluster_fraction() const { return 0.5 / std::exp(error_rate * (double)k); }

} // namespace drprg
