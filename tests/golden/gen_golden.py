#!/usr/bin/env python3
"""Generate tests/golden/vectors.json.

Three kinds of content, kept apart in the file:
  "reference": inputs and known answers TRANSCRIBED from the reference's own tests (data only: hex strings,
               secret-key byte lists, message lists, expected error variants / booleans) with file:line.
  "external":  published vectors that are not in the reference (RFC 9380 J.10.1, an Eth2 BLS sign vector).
               The reference pins neither hash_to_curve outputs nor any signature bytes, so
               these are the end-to-end pins of the hash/sign path.
  "model":     expected outputs computed by the independent big-integer Python model oracle/pymodel/bls12_381.py
               (affine formulas, literal final exponent). The C oracle and the HIP kernels are checked against them.

The reference itself (Rust + the absent amcl submodule) can be neither built nor imported here, so no vector is
produced by running it. Run from the repo root:  python3 tests/golden/gen_golden.py   (about 3 minutes)
"""
import json, os, random, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "pymodel"))
import bls12_381 as M

def hx(b): return bytes(b).hex()
def g2c(pt): return hx(M.g2_compress(pt))
def g1c(pt): return hx(M.g1_compress(pt))
def g1u(pt): return hx(M.g1_serialize_uncompressed(pt))
def skb(x): return int(x).to_bytes(32, "big")

reference = {
    "g1_compressed_round_trip": {  # reference src/amcl_utils.rs:83,89,95
        "cite": "src/amcl_utils.rs:81-99",
        "hex": ["b53d21a4cfd562c469cc81514d4ce5a6b577d8403d32a394dc265dd190b47fa9f829fdd7963afdf972e5e77854051f6f",
                "b301803f8b5ac4a1133581fc676dfedc60d891dd5fa99028805e5ea5b08d3491af75d0707adab3b70c6a6a580217bf81",
                "a491d1b0ecd9bb917989f0e74f0dea0422eac4a873e5e2644f368dffb9a6e20fd6e10c1b77654d067c0618f6e5a7f79a"]},
    "g2_compressed_round_trip": {  # reference src/amcl_utils.rs:120-121,129-130,138-139
        "cite": "src/amcl_utils.rs:118-145",
        "hex": ["a666d31d7e6561371644eb9ca7dbcb87257d8fd84a09e38a7a491ce0bbac64a324aa26385aebc99f47432970399a2ecb0def2d4be359640e6dae6438119cbdc4f18e5e4496c68a979473a72b72d3badf98464412e9d8f8d2ea9b31953bb24899",
                "a63e88274adb7a98d112c16f7057f388786496c8f57e03ee9052b46b15eb0166645008f8cc929eb4475e386f3e6f1df81181e97fac61e371a22f34a4622f7e343ca0d99846b175a92ad1bf1df6fd4d0800e4edb7c2eb3d8437ed10cbc2d88823",
                "b090fbc9d5c6c80fec73c567202a75664cd00c2592e472a4d81d2ed4b6a166311e809ca25eb88c5d0189cbf1baa8ea7918ca20f0b66678c0230e65eb4ebb3d621940984f71eb5481453e4489dafcc7f6ee2c863b76671467002a8f2392063005"]},
    "structural": {
        "pk_infinity_unchecked_ok": {"cite": "src/keys.rs:250-258", "compressed": hx(bytes([192]) + bytes(47))},
        "pk_infinity_bad_flags": {"cite": "src/keys.rs:344-350", "compressed": hx(bytes([196]) + bytes(47)), "from_bytes_err": "InvalidPoint"},
        "pk_zero_two": {"cite": "src/keys.rs:334-341", "compressed": hx(bytes([128]) + bytes(47)), "from_bytes_err": "InvalidPoint", "unchecked_ok": True},
        "pk_uncompressed_bad_sizes": {"cite": "src/keys.rs:261-273", "sizes": [1, 95, 97, 0], "fill": 1, "err": "InvalidG1Size"},
        "pk_uncompressed_off_curve": {"cite": "src/keys.rs:276-282", "uncompressed": hx(bytes(47) + b"\x01" + bytes(47) + b"\x01"), "err": "InvalidPoint"},
        "sk_errors": {"cite": "src/keys.rs:285-297",
                      "cases": [{"hex": "", "err": "InvalidSecretKeySize"}, {"hex": "01" * 33, "err": "InvalidSecretKeySize"},
                                {"hex": "00" * 32, "err": "InvalidSecretKeyRange"}, {"hex": "ff" * 32, "err": "InvalidSecretKeyRange"}]},
        "split_zero": {"cite": "src/aggregates.rs:392-410", "sk_one": hx(skb(1)),
                       "sk_minus_one": "73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000",
                       "msg": hx(bytes(32)), "fast_aggregate_verify": False},
        "empty_keys": {"cite": "src/aggregates.rs:384-389", "msg": hx(bytes(32)), "fast_aggregate_verify": False},
    },
    "readme_sk": {"cite": "src/signature.rs:105-108", "bytes": [78, 252, 122, 126, 32, 0, 75, 89, 252, 31, 42, 130, 254, 88, 6, 90, 138, 202, 135, 194, 233, 117, 181, 75, 96, 238, 79, 100, 237, 59, 140, 111], "msg": hx(b"cats")},
    "known_keys": {  # reference src/aggregates.rs:555-609
        "cite": "src/aggregates.rs:555-609",
        "control": [[40, 129, 16, 229, 203, 159, 171, 37, 94, 38, 3, 24, 17, 213, 243, 246, 122, 105, 202, 156, 186, 237, 54, 148, 116, 130, 20, 138, 15, 134, 45, 73]],
        "signing": [
            [98, 161, 50, 32, 254, 87, 16, 25, 167, 79, 192, 116, 176, 74, 164, 217, 40, 57, 179, 15, 19, 21, 240, 100, 70, 127, 111, 170, 129, 137, 42, 53],
            [53, 72, 211, 104, 184, 68, 142, 208, 115, 22, 156, 97, 28, 216, 228, 102, 4, 218, 116, 226, 166, 131, 67, 7, 40, 55, 157, 167, 157, 127, 143, 13],
            [94, 157, 163, 128, 239, 119, 116, 194, 162, 172, 189, 100, 36, 33, 13, 31, 137, 177, 80, 73, 119, 126, 246, 215, 123, 178, 195, 12, 141, 65, 65, 89],
            [74, 195, 255, 195, 62, 36, 197, 48, 100, 25, 121, 8, 191, 219, 73, 136, 227, 203, 98, 123, 204, 27, 197, 66, 193, 107, 115, 53, 5, 98, 137, 77],
            [82, 16, 65, 222, 228, 32, 47, 1, 245, 135, 169, 125, 46, 120, 57, 149, 121, 254, 168, 52, 30, 221, 150, 186, 157, 141, 25, 143, 175, 196, 21, 176]],
        "non_signing": [
            [6, 235, 126, 159, 58, 82, 170, 175, 73, 188, 251, 60, 79, 24, 164, 146, 88, 210, 177, 65, 62, 183, 124, 129, 109, 248, 181, 29, 16, 128, 207, 23],
            [100, 177, 235, 229, 217, 215, 204, 237, 178, 196, 182, 51, 28, 147, 58, 24, 79, 134, 41, 185, 153, 133, 229, 195, 32, 221, 247, 171, 91, 196, 65, 250],
            [65, 154, 236, 86, 178, 14, 179, 117, 113, 4, 40, 173, 150, 221, 23, 7, 117, 162, 173, 104, 172, 241, 111, 31, 170, 241, 185, 31, 69, 164, 115, 126],
            [13, 67, 192, 157, 69, 188, 53, 161, 77, 187, 133, 49, 254, 165, 47, 189, 185, 150, 23, 231, 143, 31, 64, 208, 134, 147, 53, 53, 228, 225, 104, 62],
            [22, 66, 26, 11, 101, 38, 37, 1, 148, 156, 162, 211, 37, 231, 37, 222, 172, 36, 224, 218, 187, 127, 122, 195, 229, 234, 124, 91, 246, 73, 12, 120]],
        "messages_hex": [hx(b"Small msg"), hx(b"cats lol"), "2a*133700"],   # src/aggregates.rs:436 ("2a*133700" = [42u8; 133700])
    },
}

external = {
    "rfc9380_J_10_1": {  # suite BLS12381G2_XMD:SHA-256_SSWU_RO_
        "dst": hx(b"QUUX-V01-CS02-with-BLS12381G2_XMD:SHA-256_SSWU_RO_"),
        "vectors": [
            {"msg": "", "x_c0": "0141ebfbdca40eb85b87142e130ab689c673cf60f1a3e98d69335266f30d9b8d4ac44c1038e9dcdd5393faf5c41fb78a",
             "x_c1": "05cb8437535e20ecffaef7752baddf98034139c38452458baeefab379ba13dff5bf5dd71b72418717047f5b0f37da03d",
             "y_c0": "0503921d7f6a12805e72940b963c0cf3471c7b2a524950ca195d11062ee75ec076daf2d4bc358c4b190c0c98064fdd92",
             "y_c1": "12424ac32561493f3fe3c260708a12b7c620e7be00099a974e259ddc7d1f6395c3c811cdd19f1e8dbf3e9ecfdcbab8d6"},
            {"msg": hx(b"abc"), "x_c0": "02c2d18e033b960562aae3cab37a27ce00d80ccd5ba4b7fe0e7a210245129dbec7780ccc7954725f4168aff2787776e6",
             "x_c1": "139cddbccdc5e91b9623efd38c49f81a6f83f175e80b06fc374de9eb4b41dfe4ca3a230ed250fbe3a2acf73a41177fd8",
             "y_c0": "1787327b68159716a37440985269cf584bcb1e621d3a7202be6ea05c4cfe244aeb197642555a0645fb87bf7466b2ba48",
             "y_c1": "00aa65dae3c8d732d10ecd2c50f8a1baf3001578f71c694e03866e9f3d49ac1e1ce70dd94a733534f106d4cec0eddd16"},
            # round 5: the remaining three messages of J.10.1
            {"msg": hx(b"abcdef0123456789"), "x_c0": "121982811d2491fde9ba7ed31ef9ca474f0e1501297f68c298e9f4c0028add35aea8bb83d53c08cfc007c1e005723cd0",
             "x_c1": "190d119345b94fbd15497bcba94ecf7db2cbfd1e1fe7da034d26cbba169fb3968288b3fafb265f9ebd380512a71c3f2c",
             "y_c0": "05571a0f8d3c08d094576981f4a3b8eda0a8e771fcdcc8ecceaf1356a6acf17574518acb506e435b639353c2e14827c8",
             "y_c1": "0bb5e7572275c567462d91807de765611490205a941a5a6af3b1691bfe596c31225d3aabdf15faff860cb4ef17c7c3be"},
            {"msg": hx(b"q128_" + b"q" * 128), "x_c0": "19a84dd7248a1066f737cc34502ee5555bd3c19f2ecdb3c7d9e24dc65d4e25e50d83f0f77105e955d78f4762d33c17da",
             "x_c1": "0934aba516a52d8ae479939a91998299c76d39cc0c035cd18813bec433f587e2d7a4fef038260eef0cef4d02aae3eb91",
             "y_c0": "14f81cd421617428bc3b9fe25afbb751d934a00493524bc4e065635b0555084dd54679df1536101b2c979c0152d09192",
             "y_c1": "09bcccfa036b4847c9950780733633f13619994394c23ff0b32fa6b795844f4a0673e20282d07bc69641cee04f5e5662"},
            {"msg": hx(b"a512_" + b"a" * 512), "x_c0": "01a6ba2f9a11fa5598b2d8ace0fbe0a0eacb65deceb476fbbcb64fd24557c2f4b18ecfc5663e54ae16a84f5ab7f62534",
             "x_c1": "11fca2ff525572795a801eed17eb12785887c7b63fb77a42be46ce4a34131d71f7a73e95fee3f812aea3de78b4d01569",
             "y_c0": "0b6798718c8aed24bc19cb27f866f1c9effcdbf92397ad6448b5c9db90d2b9da6cbabf48adc1adf59a1a28344e79d57e",
             "y_c1": "03a47f8e6d1763ba0cad63d6114c0accbef65707825a511b251a660a9b3994249ae4e63fac38b23da0c398689ee2ab52"}]},
    "eth2_sign": {  # Eth2 BLS spec test "sign_case_*": POP ciphersuite, the DST amcl's proof_of_possession::DST_G2 holds
        "sk": "263dbd792f5b1be47ed85f8938c0f29586af0d3ac7b977f21c278fe1462040e3", "msg": "00" * 32,
        "pk": "a491d1b0ecd9bb917989f0e74f0dea0422eac4a873e5e2644f368dffb9a6e20fd6e10c1b77654d067c0618f6e5a7f79a",
        "sig": "b6ed936746e01f8ecf281f020953fbf1f01debd5657c4a383940b020b26507f6076334f91e2366c96e9ab279fb5158090352ea1c5b0c9274504f4f0e7053af24802e51e4568d164fe986834f41e55c8e850ce1f98458c0cfc9ab380b55285a55"},
    # round 5: more of the Eth2 BLS spec tests (consensus-spec-tests, bls/{sign,aggregate,fast_aggregate_verify,aggregate_verify}; POP ciphersuite). Transcribed
    # from the published cases; a string is kept only where the independent big-integer model reproduces it bit for bit (build_model asserts each).
    "eth2_sign_cases": [
        {"sk": "263dbd792f5b1be47ed85f8938c0f29586af0d3ac7b977f21c278fe1462040e3", "msg": "ab" * 32,
         "sig": "91347bccf740d859038fcdcaf233eeceb2a436bcaaee9b2aa3bfb70efe29dfb2677562ccbea1c8e061fb9971b0753c240622fab78489ce96768259fc01360346da5b9f579e5da0d941e4c6ba18a0e64906082375394f337fa1af2b7127b0d121"},
        {"sk": "47b8192d77bf871b62e87859d653922725724a5c031afeabc60bcef5ff665138", "msg": "00" * 32,
         "sig": "b23c46be3a001c63ca711f87a005c200cc550b9429d5f4eb38d74322144f1b63926da3388979e5321012fb1a0526bcd100b5ef5fe72628ce4cd5e904aeaa3279527843fae5ca9ca675f4f51ed8f83bbf7155da9ecc9663100a885d5dc6df96d9"},
        {"sk": "47b8192d77bf871b62e87859d653922725724a5c031afeabc60bcef5ff665138", "msg": "56" * 32,
         "sig": "af1390c3c47acdb37131a51216da683c509fce0e954328a59f93aebda7e4ff974ba208d9a4a2a2389f892a9d418d618418dd7f7a6bc7aa0da999a9d3a5b815bc085e14fd001f6a1948768a3f4afefc8b8240dda329f984cb345c6363272ba4fe"},
        {"sk": "47b8192d77bf871b62e87859d653922725724a5c031afeabc60bcef5ff665138", "msg": "ab" * 32,
         "sig": "9674e2228034527f4c083206032b020310face156d4a4685e2fcaec2f6f3665aa635d90347b6ce124eb879266b1e801d185de36a0a289b85e9039662634f2eea1e02e670bc7ab849d006a70b2f93b84597558a05b879c8d445f387a5d5b653df"},
        {"sk": "328388aff0d4a5b7dc9205abd374e7e98f3cd9f3418edb4eafda5fb16473d216", "msg": "00" * 32,
         "sig": "948a7cb99f76d616c2c564ce9bf4a519f1bea6b0a624a02276443c245854219fabb8d4ce061d255af5330b078d5380681751aa7053da2c98bae898edc218c75f07e24d8802a17cd1f6833b71e58f5eb5b94208b4d0bb3848cecb075ea21be115"},
        {"sk": "328388aff0d4a5b7dc9205abd374e7e98f3cd9f3418edb4eafda5fb16473d216", "msg": "56" * 32,
         "sig": "a4efa926610b8bd1c8330c918b7a5e9bf374e53435ef8b7ec186abf62e1b1f65aeaaeb365677ac1d1172a1f5b44b4e6d022c252c58486c0a759fbdc7de15a756acc4d343064035667a594b4c2a6f0b0b421975977f297dba63ee2f63ffe47bb6"},
        {"sk": "328388aff0d4a5b7dc9205abd374e7e98f3cd9f3418edb4eafda5fb16473d216", "msg": "ab" * 32,
         "sig": "ae82747ddeefe4fd64cf9cedb9b04ae3e8a43420cd255e3c7cd06a8d88b7c7f8638543719981c5d16fa3527c468c25f0026704a6951bde891360c7e8d12ddee0559004ccdbe6046b55bae1b257ee97f7cdb955773d7cf29adf3ccbb9975e4eb9"}],
    # fast_aggregate_verify_valid_*: the first one, two, three of the standard keys on 00.. / 56.. / ab..; the third is also aggregate_0xabab..
    "eth2_fast_aggregate_verify": [
        {"n_keys": 1, "msg": "00" * 32,
         "sig": "b6ed936746e01f8ecf281f020953fbf1f01debd5657c4a383940b020b26507f6076334f91e2366c96e9ab279fb5158090352ea1c5b0c9274504f4f0e7053af24802e51e4568d164fe986834f41e55c8e850ce1f98458c0cfc9ab380b55285a55"},
        {"n_keys": 2, "msg": "56" * 32,
         "sig": "912c3615f69575407db9392eb21fee18fff797eeb2fbe1816366ca2a08ae574d8824dbfafb4c9eaa1cf61b63c6f9b69911f269b664c42947dd1b53ef1081926c1e82bb2a465f927124b08391a5249036146d6f3f1e17ff5f162f779746d830d1"},
        {"n_keys": 3, "msg": "ab" * 32,
         "sig": "9712c3edd73a209c742b8250759db12549b3eaf43b5ca61376d9f30e2747dbcf842d8b2ac0901d2a093713e20284a7670fcf6954e9ab93de991bb9b313e664785a075fc285806fa5224c82bde146561b446ccfc706a64b8579513cfc4ff1d930"}],
    # aggregate_verify_valid: the three standard keys on 00.. , 56.. , ab..
    "eth2_aggregate_verify": {"msgs": ["00" * 32, "56" * 32, "ab" * 32],
        "sig": "9104e74b9dfd3ad502f25d6a5ef57db0ed7d9a0e00f3500586d8ce44231212542fcfaf87840539b398bf07626705cf1105d246ca1062c6c2e1a53029a0f790ed5e3cb1f52f8234dc5144c45fc847c0cd37a92d68e7c5ba7c648a8a339f171244"},
    "eth2_sk_to_pk": [  # the three G1 strings of src/amcl_utils.rs:83-95 are [sk]G1 for the standard Eth2 test keys
        {"sk": "263dbd792f5b1be47ed85f8938c0f29586af0d3ac7b977f21c278fe1462040e3", "pk": "a491d1b0ecd9bb917989f0e74f0dea0422eac4a873e5e2644f368dffb9a6e20fd6e10c1b77654d067c0618f6e5a7f79a"},
        {"sk": "47b8192d77bf871b62e87859d653922725724a5c031afeabc60bcef5ff665138", "pk": "b301803f8b5ac4a1133581fc676dfedc60d891dd5fa99028805e5ea5b08d3491af75d0707adab3b70c6a6a580217bf81"},
        {"sk": "328388aff0d4a5b7dc9205abd374e7e98f3cd9f3418edb4eafda5fb16473d216", "pk": "b53d21a4cfd562c469cc81514d4ce5a6b577d8403d32a394dc265dd190b47fa9f829fdd7963afdf972e5e77854051f6f"}],
}

def expand_msg(h):
    return bytes([42]) * 133700 if h == "2a*133700" else bytes.fromhex(h)

def build_model():
    t0 = time.time()
    m = {}
    # ---- external pins hold for the model itself
    dst = bytes.fromhex(external["rfc9380_J_10_1"]["dst"])
    for v in external["rfc9380_J_10_1"]["vectors"]:
        p = M.hash_to_curve_g2(bytes.fromhex(v["msg"]), dst)
        assert (p[0][0], p[0][1], p[1][0], p[1][1]) == tuple(int(v[k], 16) for k in ("x_c0", "x_c1", "y_c0", "y_c1"))
    e = external["eth2_sign"]
    assert g2c(M.sign(bytes.fromhex(e["msg"]), int(e["sk"], 16))) == e["sig"]
    for kp in external["eth2_sk_to_pk"]:
        assert g1c(M.sk_to_pk(int(kp["sk"], 16))) == kp["pk"]
    for e in external["eth2_sign_cases"]:
        assert g2c(M.sign(bytes.fromhex(e["msg"]), int(e["sk"], 16))) == e["sig"]
    std = [int(kp["sk"], 16) for kp in external["eth2_sk_to_pk"]]
    for e in external["eth2_fast_aggregate_verify"]:
        agg = None
        for sk in std[:e["n_keys"]]:
            agg = M.g2_add(agg, M.sign(bytes.fromhex(e["msg"]), sk))
        assert g2c(agg) == e["sig"]
    agg = None
    for sk, mh in zip(std, external["eth2_aggregate_verify"]["msgs"]):
        agg = M.g2_add(agg, M.sign(bytes.fromhex(mh), sk))
    assert g2c(agg) == external["eth2_aggregate_verify"]["sig"]
    # ---- hash_to_curve_g2 with the POP tag
    msgs = [b"", b"a", b"an example", b"cats", b"Small msg", b"cats lol", b"Some msg", b"signed message", bytes(32), bytes(range(32)),
            bytes([1]) * 32, bytes(range(200)), bytes([42]) * 133700]
    m["hash_to_g2"] = [{"msg": ("2a*133700" if len(x) == 133700 else hx(x)), "compressed": g2c(M.hash_to_curve_g2(x))} for x in msgs]
    print("hash vectors", time.time() - t0, flush=True)
    # ---- keys
    kk = reference["known_keys"]
    sks = {"control": [int.from_bytes(bytes(b), "big") for b in kk["control"]],
           "signing": [int.from_bytes(bytes(b), "big") for b in kk["signing"]],
           "non_signing": [int.from_bytes(bytes(b), "big") for b in kk["non_signing"]]}
    readme_sk = int.from_bytes(bytes(reference["readme_sk"]["bytes"]), "big")
    pks = {k: [M.sk_to_pk(s) for s in v] for k, v in sks.items()}
    m["known_pks"] = {k: [{"compressed": g1c(p), "uncompressed": g1u(p)} for p in v] for k, v in pks.items()}
    m["readme"] = {"pk": g1c(M.sk_to_pk(readme_sk)), "sig": g2c(M.sign(b"cats", readme_sk))}
    m["minus_g1"] = g1c(M.sk_to_pk(M.R - 1))
    # ---- the comprehensive aggregate scenario (reference src/aggregates.rs:423-530) on the fixed keys
    scen = []
    for mh in kk["messages_hex"]:
        msg = expand_msg(mh)
        H = M.hash_to_curve_g2(msg)
        sig = {s: M.g2_mul(H, s) for s in sks["signing"] + sks["non_signing"][:1]}
        agg = None
        for s in sks["signing"]:
            agg = M.g2_add(agg, sig[s])
        apk = M.aggregate_pks(pks["signing"])
        neg = lambda q: M.g1_neg(q)
        def check(sg, pk):   # e(sig,-G1) e(H,pk) == 1 (subgroup membership holds by construction)
            return M.pairing_product_is_one([(sg, neg(M.G1)), (H, pk)])
        H2 = M.hash_to_curve_g2(b"different_msg!1")
        distinct = M.g2_add(M.g2_mul(H2, sks["signing"][0]), M.g2_add(agg, M.g2_neg(sig[sks["signing"][0]])))
        item = {
            "msg": mh,
            "individual_sigs": [g2c(sig[s]) for s in sks["signing"]],
            "individual_verify_own_key": [check(sig[s], p) for s, p in zip(sks["signing"][:2], pks["signing"][:2])],
            "individual_verify_control_key": [check(sig[sks["signing"][0]], pks["control"][0])],
            "agg_sig": g2c(agg), "agg_pk": g1c(apk), "agg_pk_uncompressed": g1u(apk),
            "full_set": check(agg, apk),
            "reversed_set": check(agg, M.aggregate_pks(list(reversed(pks["signing"])))),
            "double_signed_sig": g2c(M.g2_add(agg, sig[sks["signing"][0]])),
            "double_signed": check(M.g2_add(agg, sig[sks["signing"][0]]), apk),
            "distinct_msg_sig": g2c(distinct), "distinct_msg": check(distinct, apk),
            "super_set_sig": g2c(M.g2_add(agg, sig[sks["non_signing"][0]])),
            "super_set": check(M.g2_add(agg, sig[sks["non_signing"][0]]), apk),
            "subset": check(agg, M.aggregate_pks(pks["signing"][:-1])),
            "non_signing": check(agg, M.aggregate_pks(pks["non_signing"])),
        }
        scen.append(item)
        print("scenario", mh[:16], time.time() - t0, flush=True)
    m["aggregate_scenarios"] = scen
    # ---- subgroup / codec probes
    rnd = random.Random(0x6d626c73)
    probes = []
    while len(probes) < 3:      # points on E'(Fp2) outside G2: map_to_curve output before cofactor clearing
        u = (rnd.randrange(M.P), rnd.randrange(M.P))
        q = M.iso3_g2(M.sswu_g2(u))
        assert M.g2_on_curve(q) and not M.subgroup_check_g2(q)
        probes.append({"compressed": g2c(q), "in_g2": False})
    probes.append({"compressed": g2c(M.g2_mul(M.G2, 12345)), "in_g2": True})
    probes.append({"compressed": g2c(None), "in_g2": True})
    m["g2_subgroup_probes"] = probes
    g1p = []
    x = 1
    while len(g1p) < 3:         # points on E(Fp) outside G1
        y = M.fp_sqrt((x * x * x + 4) % M.P)
        if y is not None and not M.subgroup_check_g1((x, y)):
            g1p.append({"compressed": g1c((x, y)), "uncompressed": g1u((x, y)), "key_validate": False})
        x += 1
    g1p.append({"compressed": g1c(M.g1_mul(M.G1, 777)), "uncompressed": g1u(M.g1_mul(M.G1, 777)), "key_validate": True})
    m["g1_subgroup_probes"] = g1p
    bad = []
    xb = 1
    while len(bad) < 2:         # x with no point on the curve
        if M.fp_sqrt((xb ** 3 + 4) % M.P) is None:
            bb = bytearray(xb.to_bytes(48, "big")); bb[0] |= 0x80; bad.append(hx(bb))
        xb += 1
    bad.append(hx((M.P).to_bytes(48, "big")[:0] + bytes([0x80 | ((M.P >> 376) & 0x1F)]) + (M.P).to_bytes(48, "big")[1:]))   # x = p (non-canonical)
    m["g1_bad_compressed"] = bad
    # ---- verify_multiple (reference src/aggregates.rs:688-805 shape: n sets x m keys, 32-byte messages i*32)
    n, mk = 4, 2
    key_pool = [rnd.randrange(1, M.R) for _ in range(n * mk)]
    sets, sets_bad = [], []
    wrong = int.from_bytes(bytes([1]) * 32, "big")
    for i in range(n):
        msg = bytes([i]) * 32
        H = M.hash_to_curve_g2(msg)
        ks = key_pool[i * mk:(i + 1) * mk]
        apk = M.aggregate_pks([M.sk_to_pk(s) for s in ks])
        sg = M.g2_mul(H, sum(ks) % M.R)
        sgw = M.g2_mul(H, (wrong * mk) % M.R)
        sets.append((sg, apk, msg)); sets_bad.append((sgw, apk, msg))
    rands = [rnd.randrange(1, 1 << 63) for _ in range(n)]
    m["verify_multiple"] = {
        "rands": rands,
        "valid": {"sets": [{"sig": g2c(s), "apk": g1u(a), "msg": hx(mm)} for s, a, mm in sets], "result": M.verify_multiple(sets, rands)},
        "invalid": {"sets": [{"sig": g2c(s), "apk": g1u(a), "msg": hx(mm)} for s, a, mm in sets_bad], "result": M.verify_multiple(sets_bad, rands)}}
    print("verify_multiple", time.time() - t0, flush=True)
    # ---- aggregate_verify (reference src/aggregates.rs:808-929)
    na = 4
    aks = [rnd.randrange(1, M.R) for _ in range(na)]
    amsgs = [bytes([i]) * 32 for i in range(na)]
    apks = [M.sk_to_pk(s) for s in aks]
    def agg_sig(ms, ks):
        a = None
        for mm, s in zip(ms, ks):
            a = M.g2_add(a, M.sign(mm, s))
        return a
    full = agg_sig(amsgs, aks)
    rep_msgs = amsgs[:-1] + [bytes(32)]
    rep = agg_sig(rep_msgs, aks)
    missing = agg_sig(amsgs[:-1], aks[:-1])
    m["aggregate_verify"] = {
        "pks_uncompressed": [g1u(p) for p in apks],
        "valid": {"msgs": [hx(x) for x in amsgs], "sig": g2c(full), "result": M.aggregate_verify(full, amsgs, apks)},
        "msg_repeat": {"msgs": [hx(x) for x in rep_msgs], "sig": g2c(rep), "result": M.aggregate_verify(rep, rep_msgs, apks)},
        "missing_signature": {"msgs": [hx(x) for x in amsgs], "sig": g2c(missing), "result": M.aggregate_verify(missing, amsgs, apks)}}
    print("aggregate_verify", time.time() - t0, flush=True)
    # ---- a small fast_aggregate_verify batch with every rejection class (expected bitmap by the model)
    K = 3
    pool = [rnd.randrange(1, M.R) for _ in range(6)]
    ppk = [M.sk_to_pk(s) for s in pool]
    items = []
    for i in range(8):
        idx = rnd.sample(range(6), K)
        msg = bytes(rnd.getrandbits(8) for _ in range(32))
        sg = M.g2_mul(M.hash_to_curve_g2(msg), sum(pool[j] for j in idx) % M.R)
        keys = [ppk[j] for j in idx]
        kind = ["valid", "valid", "flip_msg", "wrong_key", "sig_not_in_g2", "sig_infinity", "apk_infinity", "valid"][i]
        if kind == "flip_msg": msg = bytes([msg[0] ^ 1]) + msg[1:]
        if kind == "wrong_key": keys[0] = ppk[[j for j in range(6) if j not in idx][0]]
        if kind == "sig_not_in_g2": sg = M.iso3_g2(M.sswu_g2((rnd.randrange(M.P), rnd.randrange(M.P))))
        if kind == "sig_infinity": sg = None
        if kind == "apk_infinity": keys[-1] = M.g1_neg(M.aggregate_pks(keys[:-1]))
        items.append({"kind": kind, "sig": g2c(sg), "msg": hx(msg), "pks_compressed": [g1c(k) for k in keys],
                      "pks_uncompressed": [g1u(k) for k in keys], "result": M.fast_aggregate_verify(sg, msg, keys)})
    m["fast_aggregate_verify_batch"] = {"k": K, "items": items}
    print("batch", time.time() - t0, flush=True)
    return m

if __name__ == "__main__":
    out = {"_generator": "tests/golden/gen_golden.py", "reference": reference, "external": external, "model": build_model()}
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(HERE, "vectors.json"))
