// pack.h -- host side of the 2-bit packed read format (kernels.h SketchArgs::packed; SURVEY.md section 8f NEXT-4): what crosses PCIe
// when drprg_hip_set_input_format(ctx, 1) is in force.  Letter = bits 2:1 of the byte (A 0, C 1, T 2, G 3 for either case; any other
// byte gets the letter of those bits too and its position goes to `npos`), base i of the stream in bits [2 (i & 31) + 1 : 2 (i & 31)]
// of 64-bit word i >> 5 -- read as 32-bit words that is 16 bases per word, first base lowest.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace drprg {

// Appends seq[0, len) to the packed stream that holds n_bases bases so far.  words64: room for (n_bases + len) / 32 + 2 words; the
// word that holds base n_bases (if any base of it is set) must be valid below that base and ZERO above, which is how this function
// leaves it (a fresh stream: words64[0] = 0).  Positions of bytes that are not ACGTacgt are appended to npos (as n_bases + i).
void pack_append(uint64_t* words64, uint64_t& n_bases, const char* seq, size_t len, std::vector<uint64_t>& npos);

} // namespace drprg
