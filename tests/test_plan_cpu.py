"""The routing of the verification pipeline as data (include/mbls.h: mbls_default_limits, mbls_plan_batch -- pure functions, no GPU): the table of DESIGN.md section 5 and
of tests/test_gpu_engines.py's header, checked at every boundary on the CPU. The verification entries act on the same plan (verify_pipeline calls plan_batch / plan_pass), so
what is asserted here is what the GPU runs (tests/test_gpu_engines.py compares the results at the same boundaries with the oracle)."""
import pytest

from milagro_bls_amd import _native as N

R = 65536


def one(n, L=None):
    mode, passes = N.plan_batch(n, L)
    assert mode == N.BATCH_ONE_PASS and len(passes) == 1 and passes[0]["items"] == n and passes[0]["first_item"] == 0
    return passes[0]


def test_default_limits_follow_the_round():
    L = N.default_limits(R)
    assert (L.round_items, L.coop_max_items, L.coop_hash_max_items) == (R, 5120, 3584)
    assert (L.coop_pack_min_items, L.coop_pack_max_items, L.coop_hash_pack_min_items) == (1024, 2048, 768)
    assert (L.split_max_items, L.fork_max_items, L.hash2_max_items) == (R // 2, R * 3 // 4, R * 5 // 16)
    assert (L.tracks_min_rest, L.tracks_side_max) == (3584, R // 4)
    L2 = N.default_limits(128)                      # what tests with small rounds see
    assert (L2.split_max_items, L2.fork_max_items, L2.tracks_side_max) == (64, 96, 32)


@pytest.mark.parametrize("n,pairing,message,front,ws", [
    (1, N.PAIRING_WAVE, N.MESSAGE_WAVE, N.FRONT_ALL_BESIDE, 1),
    (768, N.PAIRING_WAVE, N.MESSAGE_WAVE, N.FRONT_ALL_BESIDE, 1), (769, N.PAIRING_WAVE, N.MESSAGE_WAVE_X4, N.FRONT_ALL_BESIDE, 1),
    (1024, N.PAIRING_WAVE, N.MESSAGE_WAVE_X4, N.FRONT_ALL_BESIDE, 1), (1025, N.PAIRING_WAVE_X2, N.MESSAGE_WAVE_X4, N.FRONT_ALL_BESIDE, 1),
    (2048, N.PAIRING_WAVE_X2, N.MESSAGE_WAVE_X4, N.FRONT_ALL_BESIDE, 1), (2049, N.PAIRING_WAVE, N.MESSAGE_WAVE_X4, N.FRONT_ALL_BESIDE, 1),
    (3584, N.PAIRING_WAVE, N.MESSAGE_WAVE_X4, N.FRONT_ALL_BESIDE, 1), (3585, N.PAIRING_WAVE, N.MESSAGE_LANES2, N.FRONT_ALL_BESIDE, 2),
    (5120, N.PAIRING_WAVE, N.MESSAGE_LANES2, N.FRONT_ALL_BESIDE, 2), (5121, N.PAIRING_LANES4, N.MESSAGE_LANES2, N.FRONT_ALL_BESIDE, 2),
    (16384, N.PAIRING_LANES4, N.MESSAGE_LANES2, N.FRONT_ALL_BESIDE, 2), (16385, N.PAIRING_LANES2, N.MESSAGE_LANES2, N.FRONT_MESSAGE_BESIDE, 2),
    (20480, N.PAIRING_LANES2, N.MESSAGE_LANES2, N.FRONT_MESSAGE_BESIDE, 2), (20481, N.PAIRING_LANES2, N.MESSAGE_LANE, N.FRONT_MESSAGE_BESIDE, 2),
    (32768, N.PAIRING_LANES2, N.MESSAGE_LANE, N.FRONT_MESSAGE_BESIDE, 2), (32769, N.PAIRING_LANE, N.MESSAGE_LANE, N.FRONT_MESSAGE_BESIDE, 1),
    (49152, N.PAIRING_LANE, N.MESSAGE_LANE, N.FRONT_MESSAGE_BESIDE, 1), (49153, N.PAIRING_LANE, N.MESSAGE_LANE, N.FRONT_IN_A_ROW, 1),
    (65536, N.PAIRING_LANE, N.MESSAGE_LANE, N.FRONT_IN_A_ROW, 1), (131072, N.PAIRING_LANE, N.MESSAGE_LANE, N.FRONT_IN_A_ROW, 1),
])
def test_one_pass_routes_at_every_boundary(n, pairing, message, front, ws):
    p = one(n)
    assert (p["pairing"], p["message"], p["front"]) == (pairing, message, front), p
    assert p["workspace_items"] == ws * n
    assert p["sig_subgroup_from_miller_loop"] == (1 if n > 5120 else 0)           # on the lane kernels the subgroup verdict is read off the Miller loop


def test_batches_above_a_round():
    # a small remainder follows the round as a batch of its own (on the wave engine)
    mode, ps = N.plan_batch(R + 1)
    assert mode == N.BATCH_ROUNDS_THEN_REST and [(p["first_item"], p["items"], p["stage"], p["track"]) for p in ps] == [(0, R, 0, 0), (R, 1, 1, 0)]
    assert ps[0]["pairing"] == N.PAIRING_LANE and ps[1]["pairing"] == N.PAIRING_WAVE
    mode, ps = N.plan_batch(R + 3583)
    assert mode == N.BATCH_ROUNDS_THEN_REST
    # from 3 584 items the remainder runs BESIDE the last round: track 1, its own part of the workspace, lane pairs whatever its size (never the wave engine)
    mode, ps = N.plan_batch(R + 3584)
    assert mode == N.BATCH_ROUND_BESIDE_REST and len(ps) == 2
    a, b = ps
    assert (a["first_item"], a["items"], a["track"], a["stage"], a["pairing"], a["front"]) == (0, R, 0, 0, N.PAIRING_LANE, N.FRONT_IN_A_ROW)
    assert (b["first_item"], b["items"], b["track"], b["stage"]) == (R, 3584, 1, 0)
    assert (b["pairing"], b["message"], b["workspace_first"], b["workspace_items"]) == (N.PAIRING_LANES4, N.MESSAGE_LANES2, R, 2 * 3584)
    mode, ps = N.plan_batch(R + 16384)
    assert mode == N.BATCH_ROUND_BESIDE_REST and ps[1]["items"] == 16384 and ps[1]["pairing"] == N.PAIRING_LANES4
    # above a quarter of a round: two equal halves side by side (cut at a bitmap word), their front phases in a row above 19/32 of a round
    mode, ps = N.plan_batch(R + 16448)
    assert mode == N.BATCH_TWO_HALVES
    half = ((R + 16448) // 2 + 63) // 64 * 64
    assert [(p["first_item"], p["items"], p["track"], p["stage"], p["workspace_first"]) for p in ps] == [(0, half, 0, 0, 0), (half, R + 16448 - half, 1, 0, half)]
    assert all(p["pairing"] == N.PAIRING_LANE and p["front"] == N.FRONT_IN_A_ROW for p in ps)
    mode, ps = N.plan_batch(100000)
    assert mode == N.BATCH_TWO_HALVES and ps[0]["items"] == 50048 and ps[1]["items"] == 49952 and all(p["front"] == N.FRONT_IN_A_ROW for p in ps)
    # whole rounds in front run first, as one launch per kernel; then the last round and the remainder on two tracks
    mode, ps = N.plan_batch(150000)
    assert mode == N.BATCH_TWO_HALVES and len(ps) == 3
    assert (ps[0]["first_item"], ps[0]["items"], ps[0]["stage"]) == (0, R, 0) and ps[1]["stage"] == ps[2]["stage"] == 1
    assert ps[1]["first_item"] == R and ps[1]["items"] + ps[2]["items"] == 150000 - R and ps[2]["track"] == 1
    mode, ps = N.plan_batch(2 * R + 8192)
    assert mode == N.BATCH_ROUND_BESIDE_REST and [(p["first_item"], p["items"], p["stage"], p["track"]) for p in ps] == [(0, R, 0, 0), (R, R, 1, 0), (2 * R, 8192, 1, 1)]
    # every item is covered exactly once, workspace parts of one stage do not overlap
    for n in (R + 1, R + 3584, R + 9999, R + 20001, 2 * R - 1, 3 * R + 12345, 5 * R + 40000):
        mode, ps = N.plan_batch(n)
        at = 0
        for p in ps:
            assert p["first_item"] == at and p["items"] > 0
            at += p["items"]
        assert at == n
        for st in set(p["stage"] for p in ps):
            parts = sorted((p["workspace_first"], p["workspace_first"] + p["workspace_items"]) for p in ps if p["stage"] == st)
            assert all(parts[i][1] <= parts[i + 1][0] for i in range(len(parts) - 1))


def test_limits_change_the_plan():
    L = N.default_limits(R)
    L.tracks_min_rest = 0                       # mbls_ctx_set_tracks(ctx, 0, ..): never two tracks
    assert N.plan_batch(100000, L)[0] == N.BATCH_ROUNDS_THEN_REST
    L = N.default_limits(R); L.tracks_side_max = 0
    assert N.plan_batch(R + 8192, L)[0] == N.BATCH_TWO_HALVES
    L = N.default_limits(R); L.coop_max_items = 0; L.coop_hash_max_items = 0          # the 'lanes' engine of the GPU tests
    p = one(100, L)
    assert (p["pairing"], p["message"]) == (N.PAIRING_LANES4, N.MESSAGE_LANES2)
    L.split_max_items = 0                                                             # 'lanes2pair': the headline kernels at every size
    p = one(100, L)
    assert (p["pairing"], p["message"]) == (N.PAIRING_LANE, N.MESSAGE_LANE)
    L = N.default_limits(128); L.coop_max_items = 0; L.coop_hash_max_items = 0; L.tracks_min_rest = 1; L.tracks_side_max = 64      # small rounds, as the GPU tests set them
    mode, ps = N.plan_batch(276, L)             # remainder 20 <= a quarter of the round: beside the last round
    assert mode == N.BATCH_ROUND_BESIDE_REST and [(p["first_item"], p["items"], p["workspace_first"]) for p in ps] == [(0, 128, 0), (128, 128, 0), (256, 20, 128)]
    assert N.plan_batch(300, L)[0] == N.BATCH_TWO_HALVES          # remainder 44 > a quarter of the round: halves, whatever side_max says
    L.tracks_side_max = 0
    mode, ps = N.plan_batch(300, L)
    assert mode == N.BATCH_TWO_HALVES and [(p["first_item"], p["items"]) for p in ps] == [(0, 128), (128, 128), (256, 44)]
    mode, ps = N.plan_batch(200, L)
    assert mode == N.BATCH_TWO_HALVES and [(p["first_item"], p["items"]) for p in ps] == [(0, 128), (128, 72)]


def test_workspace_of_a_plan_counts_the_eight_lane_key_sum():
    """a pass on the wave engine with k >= 32 keys (a multiple of 8) in a uniform layout sums its keys on eight lanes per item -- n + 8 n workspace items --, and
    the entries reserve for the WHOLE plan before the first pass is queued (no pass grows the workspace under another in flight)"""
    assert N.plan_workspace_items(1, 128) == 9 and N.plan_workspace_items(100, 128) == 900
    assert N.plan_workspace_items(100, 128, split_layout=False) == 100          # ragged sets / 48-byte keys: one lane per item
    assert N.plan_workspace_items(100, 24) == 100 and N.plan_workspace_items(100, 36) == 100      # below 32 keys / not a multiple of 8
    assert N.plan_workspace_items(5120, 128) == 9 * 5120 and N.plan_workspace_items(5121, 128) == 2 * 5121
    # a round, then a remainder on the wave engine: the remainder's 9 r items may exceed the round's own
    assert N.plan_workspace_items(R + 100, 128) == R and N.plan_workspace_items(R + 3583, 128) == R
    L = N.default_limits(8192)                   # a small device: round 8 192, remainders up to 5 120 on waves -> 9 r > round
    L.tracks_min_rest = 0
    assert N.plan_batch(8192 + 2000, L)[0] == N.BATCH_ROUNDS_THEN_REST
    assert N.plan_workspace_items(8192 + 2000, 128, True, L) == 18000
    assert N.plan_workspace_items(8192 + 2000, 128, False, L) == 8192
    assert N.plan_workspace_items(0, 128) == 0
