// The reference's unit tests restated against the C++ mirror include/milagro_bls.hpp (runs on the GPU through the C ABI).
// Each block names the reference test it mirrors. Exit code 0 = all passed.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
#include "milagro_bls.hpp"
using namespace milagro_bls;
static int fails = 0;
#define CHECK(x) do { if (!(x)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #x); fails++; } } while (0)
static Bytes hex(const char* s) { Bytes b; for (size_t i = 0; s[i] && s[i + 1]; i += 2) { unsigned v; std::sscanf(s + i, "%2x", &v); b.push_back(uint8_t(v)); } return b; }
static Bytes str(const char* s) { return Bytes(s, s + std::strlen(s)); }
template <typename F> static int err_of(F f) { try { f(); } catch (const AmclError& e) { return e.kind; } return 0; }

int main() {
    // src/amcl_utils.rs:81-99: compressed G1 round trip
    const char* g1[] = {"b53d21a4cfd562c469cc81514d4ce5a6b577d8403d32a394dc265dd190b47fa9f829fdd7963afdf972e5e77854051f6f",
                        "b301803f8b5ac4a1133581fc676dfedc60d891dd5fa99028805e5ea5b08d3491af75d0707adab3b70c6a6a580217bf81",
                        "a491d1b0ecd9bb917989f0e74f0dea0422eac4a873e5e2644f368dffb9a6e20fd6e10c1b77654d067c0618f6e5a7f79a"};
    for (auto h : g1) { Bytes b = hex(h); auto pk = PublicKey::from_bytes(b); auto o = pk.as_bytes(); CHECK(Bytes(o.begin(), o.end()) == b); CHECK(pk.key_validate()); }
    // src/signature.rs:103-125 (README): fixed secret key, message "cats"
    Bytes skb = {78, 252, 122, 126, 32, 0, 75, 89, 252, 31, 42, 130, 254, 88, 6, 90, 138, 202, 135, 194, 233, 117, 181, 75, 96, 238, 79, 100, 237, 59, 140, 111};
    SecretKey sk = SecretKey::from_bytes(skb);
    CHECK(sk.as_bytes() == skb);
    PublicKey pk = PublicKey::from_secret_key(sk);
    Signature sig = Signature::new_(str("cats"), sk);
    CHECK(sig.verify(str("cats"), pk));
    auto pkb = pk.as_bytes();
    CHECK(sig.verify(str("cats"), PublicKey::from_bytes(Bytes(pkb.begin(), pkb.end()))));
    CHECK(!sig.verify(str("dogs"), pk));                                   // src/signature.rs:89-100
    auto sb = sig.as_bytes();
    CHECK(Signature::from_bytes(Bytes(sb.begin(), sb.end())) == sig);
    // src/keys.rs:261-297, :334-350: error variants
    CHECK(err_of([] { PublicKey::from_uncompressed_bytes(Bytes(95, 1)); }) == AmclError::InvalidG1Size);
    { Bytes b(96, 0); b[47] = 1; b[95] = 1; CHECK(err_of([&] { PublicKey::from_uncompressed_bytes(b); }) == AmclError::InvalidPoint); }
    { Bytes b(48, 0); b[0] = 128; CHECK(err_of([&] { PublicKey::from_bytes(b); }) == AmclError::InvalidPoint); CHECK(!PublicKey::from_bytes_unchecked(b).key_validate()); }
    { Bytes b(48, 0); b[0] = 196; CHECK(err_of([&] { PublicKey::from_bytes(b); }) == AmclError::InvalidPoint); }
    CHECK(err_of([] { SecretKey::from_bytes(Bytes(32, 0)); }) == AmclError::InvalidSecretKeyRange);
    CHECK(err_of([] { SecretKey::from_bytes(Bytes(32, 255)); }) == AmclError::InvalidSecretKeyRange);
    CHECK(err_of([] { SecretKey::from_bytes(Bytes(33, 1)); }) == AmclError::InvalidSecretKeySize);
    CHECK(err_of([] { Signature::from_bytes(Bytes(95, 0)); }) == AmclError::InvalidG2Size);
    // src/aggregates.rs:384-410: empty key list, keys summing to infinity
    AggregateSignature inf;
    CHECK(!inf.fast_aggregate_verify(Bytes(32, 0), {}));
    { Bytes one(32, 0); one[31] = 1; PublicKey p1 = PublicKey::from_secret_key(SecretKey::from_bytes(one));
      PublicKey pm = PublicKey::from_secret_key(SecretKey::from_bytes(hex("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000")));
      CHECK(!inf.fast_aggregate_verify(Bytes(32, 0), {&p1, &pm})); }
    CHECK(err_of([] { AggregatePublicKey::aggregate({}); }) == AmclError::AggregateEmptyPoints);
    // src/aggregates.rs:423-530 (shortened): aggregate of 4 signers, subset and superset must fail
    std::mt19937 gen(7);
    auto rand_sk = [&] { Bytes b(32); for (auto& v : b) v = uint8_t(gen()); b[0] &= 0x3f; b[31] |= 1; return SecretKey::from_bytes(b); };
    std::vector<Keypair> kps; for (int i = 0; i < 4; i++) { SecretKey s = rand_sk(); kps.push_back(Keypair{s, PublicKey::from_secret_key(s)}); }
    Bytes msg = str("Small msg");
    AggregateSignature agg; std::vector<const PublicKey*> pks; std::vector<PublicKey> pkv;
    for (auto& kp : kps) { agg.add(Signature::new_(msg, kp.sk)); pks.push_back(&kp.pk); pkv.push_back(kp.pk); }
    AggregatePublicKey apk = AggregatePublicKey::into_aggregate(pkv);
    CHECK(agg.fast_aggregate_verify_pre_aggregated(msg, apk));
    CHECK(agg.fast_aggregate_verify(msg, pks));
    { auto sub = pks; sub.pop_back(); CHECK(!agg.fast_aggregate_verify(msg, sub)); }
    { AggregateSignature dbl = agg; dbl.add(Signature::new_(msg, kps[0].sk)); CHECK(!dbl.fast_aggregate_verify_pre_aggregated(msg, apk)); }
    // src/aggregates.rs:808-929: aggregate_verify, distinct messages; length mismatch -> false
    { std::vector<Bytes> msgs; AggregateSignature a; for (int i = 0; i < 4; i++) { msgs.push_back(Bytes(32, uint8_t(i))); a.add(Signature::new_(msgs[i], kps[i].sk)); }
      CHECK(a.aggregate_verify(msgs, pks));
      // the batched form: the same item three times -- as is, with a message flipped, with one message missing (:131-133) -- and an item without pairs
      { auto bad = msgs; bad[2][0] ^= 1; auto shortm = msgs; shortm.pop_back();
        auto r = aggregate_verify_batch({a, a, a, a}, {msgs, bad, shortm, {}}, {pks, pks, pks, {}});
        CHECK(r.size() == 4 && r[0] && !r[1] && !r[2] && !r[3]); }
      msgs.pop_back(); CHECK(!a.aggregate_verify(msgs, pks)); }
    // src/aggregates.rs:688-805: verify_multiple_aggregate_signatures, valid then one set signed with the wrong key
    { std::vector<AggregateSignature> sigs(3); std::vector<AggregatePublicKey> apks; std::vector<std::tuple<const AggregateSignature*, const AggregatePublicKey*, Bytes>> sets;
      for (int i = 0; i < 3; i++) { Bytes m(32, uint8_t(i)); sigs[i].add(Signature::new_(m, kps[i].sk)); apks.push_back(AggregatePublicKey::from_public_key(kps[i].pk)); }
      for (int i = 0; i < 3; i++) sets.emplace_back(&sigs[i], &apks[i], Bytes(32, uint8_t(i)));
      auto rng = [&] { return uint8_t(gen()); };
      CHECK(AggregateSignature::verify_multiple_aggregate_signatures(rng, sets));
      // the same sets cut into one shard per context of a two-context handle
      int devs[2] = {0, 0}; mbls_multi* m2 = nullptr; CHECK(mbls_multi_create(&m2, devs, 2) == MBLS_OK);
      CHECK(AggregateSignature::verify_multiple_aggregate_signatures(m2, rng, sets));
      AggregateSignature wrong; wrong.add(Signature::new_(Bytes(32, 1), kps[3].sk)); std::get<0>(sets[1]) = &wrong;
      CHECK(!AggregateSignature::verify_multiple_aggregate_signatures(rng, sets));
      CHECK(!AggregateSignature::verify_multiple_aggregate_signatures(m2, rng, sets));
      // src/aggregates.rs:272-287: scalars are drawn in the reference's order -- set i's subgroup test comes before rand[i], the first signature outside G2 ends the
      // loop: with such a signature (tests/golden/vectors.json, model.g2_subgroup_probes[0]: on the curve, not in G2) at position 1 exactly 8 bytes are consumed
      AggregateSignature outside = AggregateSignature::from_bytes(hex("b45fa214ab17cf53c091f28b93366978b5bc9e5251694f40655cede2738b21a47dcee1c2a45074b93ed0f03b5257eba201d88a9447cba04689cc1c5677a33b102be41a5a1d59ad97b755f61ee540409b4df949219cbbec5e826e63c151cf2a8a"));
      int drawn = 0; auto counting = [&] { drawn++; return uint8_t(gen()); };
      std::get<0>(sets[1]) = &sigs[1];
      CHECK(AggregateSignature::verify_multiple_aggregate_signatures(counting, sets) && drawn == 24);
      std::get<0>(sets[1]) = &outside; drawn = 0;
      CHECK(!AggregateSignature::verify_multiple_aggregate_signatures(counting, sets) && drawn == 8);
      drawn = 0; CHECK(!AggregateSignature::verify_multiple_aggregate_signatures(m2, counting, sets) && drawn == 8);
      mbls_multi_destroy(m2); }
    // src/aggregates.rs:100-106 AggregateSignature::aggregate (one batched launch) == repeated add; src/keys.rs:36-77 key generation
    { std::vector<Signature> ss; std::vector<const Signature*> ps; for (auto& kp : kps) ss.push_back(Signature::new_(msg, kp.sk)); for (auto& x : ss) ps.push_back(&x);
      CHECK(AggregateSignature::aggregate(ps) == agg); CHECK(AggregateSignature::aggregate({}) == AggregateSignature()); }
    { Bytes ikm(32); for (int i = 0; i < 32; i++) ikm[i] = uint8_t(i);
      SecretKey g = SecretKey::key_generate(ikm);
      CHECK(g.as_bytes() == hex("23360db7e337b0a32b264e06bc11c1b474d16f55665373de1ce93cf15ddb3456"));       // = hashlib/hmac restatement (tests/test_cpp_api.py)
      CHECK(err_of([] { SecretKey::key_generate(Bytes(31, 1)); }) == AmclError::InvalidSecretKeySize);
      Keypair kp = Keypair::random([&] { return uint8_t(gen()); });
      Bytes m2 = str("keygen"); CHECK(Signature::new_(m2, kp.sk).verify(m2, kp.pk)); }
    // one context shared by threads (every ABI entry takes its lock): concurrent verifications keep their own answers
    { std::vector<std::thread> th; std::vector<int> bad(4, 0);
      for (int t = 0; t < 4; t++) th.emplace_back([&, t] { for (int r = 0; r < 3; r++) { Bytes m(32, uint8_t(40 + t)); Signature s = Signature::new_(m, kps[t].sk);
          if (!s.verify(m, kps[t].pk) || s.verify(m, kps[(t + 1) % 4].pk)) bad[t]++; } });
      for (auto& x : th) x.join();
      for (int b : bad) CHECK(b == 0); }
    std::printf(fails ? "%d checks failed\n" : "all C++ API checks passed\n", fails);
    return fails ? 1 : 0;
}
