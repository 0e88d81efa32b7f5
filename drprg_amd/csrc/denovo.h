// denovo.h -- novel variants inside the candidate regions of `discover`, and the PRG update they lead to.  See denovo.cpp.
#pragma once
#include "genotype.h"
#include <functional>

namespace drprg {

struct NovelVariant {
    std::string chrom;
    uint32_t prg = 0;
    uint32_t pos = 0; // 0-based on the locus' called consensus
    std::string ref, alt; // either may be empty (insertion / deletion)
    uint32_t support = 0, spanning = 0; // reads that spell alt between the anchors / reads that hold both anchors (assembled alleles: k-mer counts)
    uint32_t group = ~0u; // variants with the same group (the candidate region they come from) are ALTERNATIVE alleles, never one haplotype
};

// second pass over the reads file (host threads): exact-anchor pile-up over every candidate region of `gr`; accurate_reads:
// whole strings are counted (Illumina), otherwise the strings are aligned to the consensus and counted column by column
// resident (may be empty): called with the packed anchor k-mers and their length instead of reading the file; it appends every
// read that holds one of them (more reads do no harm) to bases / offsets -- Mapper::select_reads_with_anchors.
using ResidentReads = std::function<void(const std::vector<uint64_t>& anchors, uint32_t anchor_len, std::vector<uint8_t>& bases, std::vector<uint64_t>& offsets)>;
std::vector<NovelVariant> assemble_candidate_regions(const GenotypeResult& gr, const std::string& reads_path, int threads, const DiscoverParams& dp,
    bool accurate_reads, const ResidentReads& resident = ResidentReads());

// <dir>/denovo_paths.txt (+ denovo_sequences.fa, denovo_variants.tsv).  list_loci = false keeps the "0 loci" line: the
// variants are then reported in denovo_variants.tsv only and the caller's make_prg step is not triggered.
void write_denovo_paths(const std::string& dir, const std::string& sample, const GenotypeResult& gr, const std::vector<NovelVariant>& variants,
    bool list_loci);

// prgs: (name, PRG string) per locus in file order; every variant becomes a new site of its locus' PRG string whose first allele is the
// stretch of the string it touches (whole sites it runs into included) and whose second allele is the called path over that stretch with
// the variant applied; markers are numbered again in pandora's parse order.  Returns the number applied; `skipped` (may be null)
// receives locus:pos of the variants that could not be placed.
uint32_t update_prgs(std::vector<std::pair<std::string, std::string>>& prgs, const GenotypeResult& gr, const std::vector<NovelVariant>& variants,
    std::vector<std::string>* skipped);

// A denovo_paths.txt (pandora discover's, or write_denovo_paths' above; layout of /root/reference/src/lib.rs:3010-3038) read back into
// the called paths and variants update_prgs takes.  names: the loci of the PRG file in order.  Throws Error(DRPRG_EINVAL) on a file
// that does not parse or names an unknown locus.
void read_denovo_paths(const std::string& path, const std::vector<std::string>& names, GenotypeResult& gr, std::vector<NovelVariant>& variants);

} // namespace drprg
