// Header-only C++ mirror of the reference's decoder classes over the C ABI (include/ldpc_amd.h).
//
//   LDPC::LogDecoder::decode(encodedH, r, e, maxL) -> DecodeResult{decodedCode, iterationNum, syndrome}
//
// keeps the names, argument meaning and result record of
//   public static DecodeResult decode(int[][] encodedH, int[] r, double e, int maxL)
// (src/LDPC/LogDecoder.java:15-17; LogDecoderBefore.java:15-17; MinSumDecoder.java:9-11).
// A per-frame call is a batch of one; decodeFrames is the batch form a driver should use instead
// of the per-frame loop of LDPC_GallagerIntStreamSimu.java:99-184.  Errors of the C ABI become
// std::runtime_error; a word of the wrong length is std::out_of_range, where the reference would
// throw ArrayIndexOutOfBoundsException.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ldpc_amd.h"

namespace LDPC {

struct DecodeResult {
    std::vector<int> decodedCode;
    int iterationNum;
    int syndrome;
};

namespace detail {

struct CtxDeleter {
    void operator()(ldpc_amd_ctx* c) const { ldpc_amd_destroy(c); }
};
using CtxPtr = std::unique_ptr<ldpc_amd_ctx, CtxDeleter>;

inline void check(int rc, const ldpc_amd_ctx* c) {
    if (rc != LDPC_AMD_OK) throw std::runtime_error(std::string("ldpc_amd: ") + ldpc_amd_last_error(c));
}

// one context per distinct matrix, kept for the life of the process (the reference rebuilds
// A(i)/B(j) on every call, LogDecoder.java:27-42)
inline ldpc_amd_ctx* context_for(const std::vector<std::vector<int>>& H, int cn_rule) {
    static std::map<std::pair<std::vector<std::vector<int>>, int>, CtxPtr> cache;
    auto key = std::make_pair(H, cn_rule);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second.get();
    std::vector<int32_t> rows, cols;
    const int m = (int)H.size(), n = m ? (int)H[0].size() : 0;
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j)
            if (H[i][j] == 1) { rows.push_back(i); cols.push_back(j); }
    ldpc_amd_ctx* c = nullptr;
    check(ldpc_amd_create(m, n, (int64_t)rows.size(), rows.data(), cols.data(), -1, &c), nullptr);
    CtxPtr holder(c);
    check(ldpc_amd_set_cn_rule(c, cn_rule), c);
    cache.emplace(std::move(key), std::move(holder));
    return c;
}

template <int LLR_MODE, int CN_RULE>
struct Decoder {
    // r: frames x n, 0/1
    static std::vector<DecodeResult> decodeFrames(const std::vector<std::vector<int>>& encodedH,
                                                  const std::vector<std::vector<int>>& r, double e, int maxL) {
        ldpc_amd_ctx* c = context_for(encodedH, CN_RULE);
        int32_t m = 0, n = 0;
        check(ldpc_amd_info(c, &m, &n, nullptr, nullptr, nullptr), c);
        const int64_t frames = (int64_t)r.size();
        std::vector<uint8_t> in((size_t)frames * n), hard((size_t)frames * n);
        std::vector<int32_t> it((size_t)frames), sw((size_t)frames);
        for (int64_t f = 0; f < frames; ++f) {
            if ((int)r[(size_t)f].size() != n) throw std::out_of_range("received word length != n");
            for (int j = 0; j < n; ++j) in[(size_t)f * n + j] = (uint8_t)(r[(size_t)f][j] != 0);
        }
        check(ldpc_amd_decode_bsc_batch(c, in.data(), frames, e, LLR_MODE, maxL, 1, hard.data(), it.data(), sw.data()), c);
        std::vector<DecodeResult> out((size_t)frames);
        for (int64_t f = 0; f < frames; ++f) {
            out[(size_t)f].decodedCode.assign(hard.begin() + f * n, hard.begin() + (f + 1) * n);
            out[(size_t)f].iterationNum = it[(size_t)f];
            out[(size_t)f].syndrome = sw[(size_t)f];
        }
        return out;
    }
    static DecodeResult decode(const std::vector<std::vector<int>>& encodedH, const std::vector<int>& r, double e,
                               int maxL) {
        return decodeFrames(encodedH, {r}, e, maxL)[0];
    }
};

}  // namespace detail

// sum-product, parity positions +-ln(1e20): LogDecoder.java
using LogDecoder = detail::Decoder<LDPC_AMD_LLR_PARITY_LN1E20, LDPC_AMD_CN_SUMPRODUCT>;
// sum-product, every position +-ln((1-e)/e): LogDecoderBefore.java
using LogDecoderBefore = detail::Decoder<LDPC_AMD_LLR_UNIFORM, LDPC_AMD_CN_SUMPRODUCT>;
// normalised min-sum, parity positions +-ln((1-1e-7)/1e-7): MinSumDecoder.java
using MinSumDecoder = detail::Decoder<LDPC_AMD_LLR_PARITY_1EM7, LDPC_AMD_CN_MINSUM>;

}  // namespace LDPC
