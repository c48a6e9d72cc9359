// Minimal C++ caller of the host mirror; mirrors the per-frame call of
// LDPC_GallagerIntStreamSimu.java:122.
//   g++ -std=c++17 ldpc-simulator_amd/host/example_decode.cpp -L ldpc-simulator_amd -lldpc_amd
//       -Wl,-rpath,$PWD/ldpc-simulator_amd -o /tmp/example_decode      (one command line)
#include <cstdio>

#include "LogDecoder.hpp"

int main() {
    const std::vector<std::vector<int>> H = {{1, 1, 1}};
    try {
        const LDPC::DecodeResult res = LDPC::LogDecoder::decode(H, {0, 0, 0}, 0.1, 5);
        std::printf("decodedCode = %d%d%d  iterationNum = %d  syndrome = %d\n", res.decodedCode[0], res.decodedCode[1],
                    res.decodedCode[2], res.iterationNum, res.syndrome);
        return (res.iterationNum == 1 && res.syndrome == 0) ? 0 : 1;
    } catch (const std::exception& ex) {
        std::printf("error: %s\n", ex.what());
        return 2;
    }
}
