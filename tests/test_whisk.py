"""The reference's `whisk` package on this stack (go-curdleproofs_amd/host/whisk.cpp), mirroring
/root/reference/whisk/whisk_test.go.  The tracker (opening) proofs need no MSM and run on the
CPU; the shuffle proof and the full lifecycle run the curdleproof prover / verifier with every
MSM on the GPU."""
import numpy as np
import pytest


def fr_limbs(oracle, k):
    return np.array(oracle.fr_to_mont_limbs(k), dtype=np.uint64)


def fr_int(oracle, limbs):
    return oracle.fr_from_mont_limbs([int(v) for v in limbs])


def compute_tracker(oracle, k, r):
    # whisk_test.go:98-104: rG = r*G, krG = k*rG, both in gnark's compressed form
    rG = oracle.scalar_mul(r, oracle.G1)
    return oracle.compress(rG) + oracle.compress(oracle.scalar_mul(k, rG))


def k_comm(oracle, k):
    return oracle.compress(oracle.scalar_mul(k, oracle.G1))  # whisk_test.go:106-109


def test_tracker_proof(cm, oracle):
    # TestWhiskTrackerProof, whisk_test.go:13-34
    rand = cm.Rand(0)
    k = fr_int(oracle, rand.get_fr())
    tracker = compute_tracker(oracle, k, fr_int(oracle, rand.get_fr()))
    proof = cm.whisk_generate_tracker_proof(tracker, fr_limbs(oracle, k), rand)
    assert len(proof) == cm.WHISK_TRACKER_PROOF_SIZE == 128
    assert cm.whisk_is_valid_tracker_proof(tracker, k_comm(oracle, k), proof) is True
    # the proof binds k: another commitment, another tracker or a touched response are rejected
    assert cm.whisk_is_valid_tracker_proof(tracker, k_comm(oracle, k + 1), proof) is False
    other = compute_tracker(oracle, k + 1, 12345)
    assert cm.whisk_is_valid_tracker_proof(other, k_comm(oracle, k), proof) is False
    touched = bytearray(proof)
    touched[127] ^= 1
    assert cm.whisk_is_valid_tracker_proof(tracker, k_comm(oracle, k), bytes(touched)) is False
    # (false, err): bytes that are not a compressed curve point / not in the subgroup / not a canonical scalar
    with pytest.raises(cm.CurdleError) as e:
        cm.whisk_is_valid_tracker_proof(tracker, k_comm(oracle, k), b"\x00" * 128)
    assert "decoding proof" in e.value.msg
    bad_s = proof[:96] + b"\xff" * 32
    with pytest.raises(cm.CurdleError) as e:
        cm.whisk_is_valid_tracker_proof(tracker, k_comm(oracle, k), bad_s)
    assert "decoding proof" in e.value.msg
    with pytest.raises(cm.CurdleError) as e:
        cm.whisk_is_valid_tracker_proof(b"\x01" * 96, k_comm(oracle, k), proof)
    assert "deserializing rG and krG" in e.value.msg


def test_tracker_proof_matches_the_protocol_equations(cm, oracle):
    """A = b*G, B = b*rG, s = b - c*k (whisk.go:156-172): recompute the verifier's two
    equations with the oracle from the decoded proof."""
    rand = cm.Rand(3)
    k, r = fr_int(oracle, rand.get_fr()), fr_int(oracle, rand.get_fr())
    tracker = compute_tracker(oracle, k, r)
    proof = cm.whisk_generate_tracker_proof(tracker, fr_limbs(oracle, k), cm.Rand(9))
    b = oracle.Rand(9).get_fr()                       # the blinder is the first draw of the proof's Rand
    assert proof[:48] == oracle.compress(oracle.scalar_mul(b, oracle.G1))
    rG = oracle.scalar_mul(r, oracle.G1)
    assert proof[48:96] == oracle.compress(oracle.scalar_mul(b, rG))
    s = int.from_bytes(proof[96:], "big")
    c = (b - s) * pow(k, -1, oracle.R) % oracle.R     # the challenge implied by s
    assert oracle.add(oracle.scalar_mul(s, oracle.G1), oracle.scalar_mul(c * k % oracle.R, oracle.G1)) == oracle.scalar_mul(b, oracle.G1)


def shuffle_trackers(cm, oracle, rand, n):
    # generateShuffleTrackers, whisk_test.go:111-119 (k then r drawn per tracker)
    out = []
    for _ in range(n):
        k = fr_int(oracle, rand.get_fr())
        r = fr_int(oracle, rand.get_fr())
        out.append(compute_tracker(oracle, k, r))
    return out


@pytest.mark.gpu
def test_shuffle_proof(gpu, oracle):
    # TestWhiskShuffleProof, whisk_test.go:36-56
    rand = gpu.Rand(0)
    crs = gpu.CRS(gpu.WHISK_ELL, rand)
    pre = shuffle_trackers(gpu, oracle, rand, gpu.WHISK_ELL)
    post, proof = gpu.whisk_generate_shuffle_proof(crs, pre, rand)
    assert len(proof) == gpu.WHISK_SHUFFLE_PROOF_SIZE == 4576 and len(post) == gpu.WHISK_ELL
    assert proof[4536:] == b"\x00" * 40            # M (48) + curdleproof (4,488), zero padded (types.go:66-68)
    assert gpu.whisk_is_valid_shuffle_proof(crs, pre, post, proof, rand) is True
    # the post trackers are a permutation of k * pre trackers: none equals its pre tracker
    assert all(a != b for a, b in zip(pre, post))
    # rejected: trackers swapped, a post tracker replaced, proof body touched
    assert gpu.whisk_is_valid_shuffle_proof(crs, post, pre, proof, gpu.Rand(1)) is False
    post2 = list(post)
    post2[5] = post[6]
    assert gpu.whisk_is_valid_shuffle_proof(crs, pre, post2, proof, gpu.Rand(1)) is False
    # (false, err) legs
    with pytest.raises(gpu.CurdleError) as e:
        gpu.whisk_is_valid_shuffle_proof(crs, pre, post[:-1], proof, gpu.Rand(1))
    assert "same length" in e.value.msg
    with pytest.raises(gpu.CurdleError) as e:
        gpu.whisk_is_valid_shuffle_proof(crs, pre, post, b"\x00" * 4576, gpu.Rand(1))
    assert "decoding proof" in e.value.msg
    bad = list(pre)
    bad[0] = b"\x01" * 96
    with pytest.raises(gpu.CurdleError) as e:
        gpu.whisk_is_valid_shuffle_proof(crs, bad, post, proof, gpu.Rand(1))
    assert "getting pre shuffle points" in e.value.msg
    # a tracker point on the curve but outside the prime-order subgroup: SetBytes rejects it
    # (types.go:85-95).  Here that verdict comes from the subgroup test that runs on the GPU
    # while the host verifies, and must still surface as the decoding error.
    p = oracle.P
    x = 6
    while True:
        rhs = (x * x * x + 4) % p
        y = pow(rhs, (p + 1) // 4, p)
        if y * y % p == rhs and oracle.scalar_mul(oracle.R, (x, y)) is not None:
            break
        x += 1
    rogue = oracle.compress((x, y))
    for which, msg in ((0, "getting pre shuffle points"), (1, "getting post shuffle points")):
        trackers = [list(pre), list(post)]
        trackers[which][7] = trackers[which][7][:48] + rogue        # krG replaced
        with pytest.raises(gpu.CurdleError) as e:
            gpu.whisk_is_valid_shuffle_proof(crs, trackers[0], trackers[1], proof, gpu.Rand(1))
        assert msg in e.value.msg and "krG" in e.value.msg
    # ... and a proof point outside the subgroup (M is the first 48 bytes of the proof)
    with pytest.raises(gpu.CurdleError) as e:
        gpu.whisk_is_valid_shuffle_proof(crs, pre, post, rogue + proof[48:], gpu.Rand(1))
    assert "decoding proof" in e.value.msg
    # the prover decodes its 2n tracker points in one GPU batch too: the same errors as getPoints
    for bad_tracker, what in ((b"\x01" * 96, "rG"), (pre[3][:48] + rogue, "krG")):
        bad = list(pre)
        bad[3] = bad_tracker
        with pytest.raises(gpu.CurdleError) as e:
            gpu.whisk_generate_shuffle_proof(crs, bad, gpu.Rand(2))
        assert "getting points" in e.value.msg and what in e.value.msg


@pytest.mark.gpu
@pytest.mark.timeout(180)
def test_more_concurrent_verifications_than_workspace_slots(gpu, oracle):
    """Each verification keeps a decode context while its subgroup test runs on the GPU and
    takes an MSM workspace slot for its final check: twelve at once (more than either pool
    holds) must all finish."""
    import threading
    rand = gpu.Rand(2)
    crs = gpu.CRS(gpu.WHISK_ELL, rand)
    pre = shuffle_trackers(gpu, oracle, rand, gpu.WHISK_ELL)
    post, proof = gpu.whisk_generate_shuffle_proof(crs, pre, rand)
    results = [None] * 12

    def worker(t):
        results[t] = all(gpu.whisk_is_valid_shuffle_proof(crs, pre, post, proof, gpu.Rand(50 + 7 * t + i)) for i in range(3))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert results == [True] * 12


@pytest.mark.gpu
def test_shuffle_proof_batch(gpu, oracle):
    """curdle_whisk_is_valid_shuffle_proof_batch: several shuffles over one CRS, all points
    decoded by one kernel, exact per-proof bits when some are bad."""
    rand = gpu.Rand(4)
    crs = gpu.CRS(gpu.WHISK_ELL, rand)
    sets = []
    for j in range(3):
        pre = shuffle_trackers(gpu, oracle, gpu.Rand(30 + j), gpu.WHISK_ELL)
        post, proof = gpu.whisk_generate_shuffle_proof(crs, pre, gpu.Rand(60 + j))
        sets.append((pre, post, proof))
    pres, posts, proofs = ([s[c] for s in sets] for c in range(3))
    assert gpu.whisk_is_valid_shuffle_proof_batch(crs, pres, posts, proofs, gpu.Rand(1), nthreads=2) == [True] * 3
    assert gpu.whisk_is_valid_shuffle_proof_batch(crs, [], [], [], gpu.Rand(1)) == []
    # proof 0 against proof 1's post trackers; proof 2 with a tracker point that is not a curve point
    bad_posts = [posts[1], posts[1], list(posts[2])]
    bad_posts[2][3] = b"\x01" * 96
    assert gpu.whisk_is_valid_shuffle_proof_batch(crs, pres, bad_posts, proofs, gpu.Rand(1), nthreads=3) == [False, True, False]
    # a proof that does not parse
    assert gpu.whisk_is_valid_shuffle_proof_batch(crs, pres[:2], posts[:2], [proofs[0], b"\x00" * 4576], gpu.Rand(1)) == [True, False]


@pytest.mark.gpu
def test_full_lifecycle(gpu, oracle):
    # TestWhiskFullLifecycle, whisk_test.go:58-91 with produceBlock / processBlock (:137-209)
    rand = gpu.Rand(0)
    crs = gpu.CRS(gpu.WHISK_ELL, rand)
    g1_gen_bytes = oracle.compress(oracle.G1)
    proposer_index = 15400
    state = {"tracker": compute_tracker(oracle, proposer_index, 1), "k_comm": k_comm(oracle, proposer_index),
             "shuffled": shuffle_trackers(gpu, oracle, rand, gpu.WHISK_ELL)}
    proposer_k = fr_int(oracle, rand.get_fr())

    def produce_block():
        r = gpu.Rand(0)
        post, shuffle_proof = gpu.whisk_generate_shuffle_proof(crs, state["shuffled"], r)
        first = state["tracker"][:48] == g1_gen_bytes
        if first:
            tracker = compute_tracker(oracle, proposer_k, fr_int(oracle, r.get_fr()))
            kc = k_comm(oracle, proposer_k)
            registration = gpu.whisk_generate_tracker_proof(tracker, fr_limbs(oracle, proposer_k), r)
        else:
            tracker, kc, registration = compute_tracker(oracle, 1, 1), k_comm(oracle, 1), b"\x00" * 128
        k_prev = proposer_index if first else proposer_k
        opening = gpu.whisk_generate_tracker_proof(state["tracker"], fr_limbs(oracle, k_prev), r)
        return {"opening": opening, "post": post, "shuffle_proof": shuffle_proof, "registration": registration,
                "tracker": tracker, "k_comm": kc}

    def process_block(block):
        r = gpu.Rand(0)
        assert gpu.whisk_is_valid_tracker_proof(state["tracker"], state["k_comm"], block["opening"]) is True
        assert gpu.whisk_is_valid_shuffle_proof(crs, state["shuffled"], block["post"], block["shuffle_proof"], r) is True
        if state["tracker"][:48] == g1_gen_bytes:
            assert gpu.whisk_is_valid_tracker_proof(block["tracker"], block["k_comm"], block["registration"]) is True
            state["tracker"], state["k_comm"] = block["tracker"], block["k_comm"]

    process_block(produce_block())   # first proposal: registers the validator's tracker
    assert state["tracker"][:48] != g1_gen_bytes
    process_block(produce_block())   # second proposal: opens the registered tracker
