// Host-only check of SecretKey::key_generate in include/milagro_bls.hpp (no GPU, no context): prints the derived secret keys
// for the (ikm, key_info) pairs given as hex on the command line; tests/test_cpp_api.py compares them with hashlib/hmac.
#include <cstdio>
#include <string>
#include "milagro_bls.hpp"
static milagro_bls::Bytes unhex(const std::string& h) { milagro_bls::Bytes b; for (size_t i = 0; i + 1 < h.size(); i += 2) b.push_back(uint8_t(std::stoi(h.substr(i, 2), nullptr, 16))); return b; }
int main(int argc, char** argv) {
    if (argc == 6 && std::string(argv[1]) == "hkdf") {        // hkdf <ikm> <salt> <info> <L>: the primitives alone (RFC 5869 vectors)
        auto dash = [](const char* s) { return std::string(s) == "-" ? std::string() : std::string(s); };
        auto prk = milagro_bls::detail::hkdf_extract(unhex(dash(argv[3])), unhex(dash(argv[2])));
        auto okm = milagro_bls::detail::hkdf_expand(prk, unhex(dash(argv[4])), size_t(std::stoi(argv[5])));
        for (uint8_t v : prk) printf("%02x", v);
        printf("\n");
        for (uint8_t v : okm) printf("%02x", v);
        printf("\n");
        return 0;
    }
    for (int i = 1; i + 1 < argc; i += 2) {
        try {
            auto sk = milagro_bls::SecretKey::key_generate(unhex(argv[i]), unhex(std::string(argv[i + 1]) == "-" ? "" : argv[i + 1]));
            for (uint8_t v : sk.as_bytes()) printf("%02x", v);
            printf("\n");
        } catch (const milagro_bls::AmclError& e) { printf("AmclError %d\n", int(e.kind)); }
    }
    return 0;
}
