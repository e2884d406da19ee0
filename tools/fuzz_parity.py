"""Randomised parity run: commit() and commit_and_generate_proof() on random blob sizes and FRI configurations, single calls and
batches, byte-compared with the CPU oracle.  Not part of the test suite (run time is whatever you ask for).
usage: python tools/fuzz_parity.py [seconds] [seed]"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import frieda_amd
from oracle import oracle as O
from conftest import splitmix64_bytes



def run(budget, seed, ctx=None, big_every=12, multi_every=9):
  """-> (single proofs, batches, seconds, MB-scale proofs).  Every `big_every`-th case (the first included) is a 0.4 - 8 MB blob:
  L = 16 .. 20/21 with ragged lengths, i.e. the strided passes of the encode's planner (padded 8-layer, generic, two-pass); every
  `multi_every`-th also sends a mixed list of blobs through frieda_prove_many / frieda_commit_many (counted among the batches)."""
  rng = random.Random(seed)
  n_big = it = 0
  ctx = ctx or frieda_amd.Context(0)
  mc = None  # the multi-GPU entry on one device (frieda_prove_many / frieda_commit_many: host blobs, upload ring, units of equal lengths)
  t0 = time.time()
  n_single = n_batch = 0
  t_print = t0
  while time.time() - t0 < budget:
      if time.time() - t_print > 30:  # a progress line every half minute (long runs under gpurun must not look hung)
          t_print = time.time()
          print(f"  ... {n_single} proofs, {n_batch} batches after {t_print - t0:.0f} s", flush=True)
      B = rng.choice([0, 0, 1, 2, 3, 4, 4, 4, 5, 6, 7, 8])  # src/lib.rs:31 takes any u32: 0 (no zero-padded layer) .. 256-fold
      big = big_every > 0 and it % big_every == 0
      it += 1
      size = rng.randint(400000, 8000000) if big else rng.choice([rng.randint(0, 300), rng.randint(300, 20000), rng.randint(20000, 400000 >> max(B - 4, 0))])
      data = splitmix64_bytes(rng.randint(1, 1 << 30), max(size, 1)).tobytes()[:size]
      # shape: F felts -> padded to a power of two >= 4 -> L = log2 - 2 -> n = L + B
      F = (8 * size + 29) // 30
      Fp = 4
      while Fp < F:
          Fp *= 2
      L = Fp.bit_length() - 1 - 2
      if big:  # keep the oracle's share of a short run bounded: domains of at most 2^22 points (~6 s of CPU per proof)
          B = rng.choice([b for b in (0, 1, 2, 4) if L + b <= 22])
      n = L + B
      if n < 1:  # Coset::half_odds(L + B - 1) underflows: the reference panics, both sides must say so
          for fn, exc in ((lambda: O.commit(data, B), RuntimeError), (lambda: ctx.commit(data, B), frieda_amd.FriedaPanic)):
              try:
                  fn()
                  raise AssertionError(("commit did not report the reference's panic", size, B))
              except exc:
                  pass
          continue
      root = ctx.commit(data, B)
      if not big:  # (a big case compares the root through the proof below: one oracle run instead of two)
          assert root == O.commit(data, B), ("commit", size, B)
      if L < 1 or n < 2:
          continue
      last = rng.randint(0, min(3, L - 1))
      nq = rng.choice([1, 2, 5, 20, 20, 33, 64, 65, 100])
      pow_bits = rng.choice([0, 3, 8, 12])
      seed = rng.choice([None, rng.randint(0, 1 << 62)])
      cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(B, last, nq), pow_bits)
      ocfg = O.make_config(pow_bits, B, last, nq)
      try:
          o_root, o_proof = O.commit_and_generate_proof(data, seed, ocfg)
      except Exception as e:  # configurations the reference panics on: the product must report the same
          try:
              ctx.commit_and_generate_proof(data, seed, cfg)
          except frieda_amd.FriedaPanic:
              continue
          raise AssertionError(("oracle panicked, product did not", size, B, last, nq, str(e)))
      r, p = ctx.commit_and_generate_proof(data, seed, cfg)
      assert r == o_root == root and p.serialize() == o_proof.serialize(), ("prove", size, B, last, nq, pow_bits, seed)
      n_single += 1
      n_big += big
      if multi_every > 0 and it % multi_every == 0:
          # a mixed list through the C ABI's multi entry: runs of equal lengths (batched units of up to four) between ragged ones
          if mc is None:
              mc = frieda_amd.MultiContext([0])
          lens = []
          for _ in range(rng.randint(2, 4)):
              ln = rng.choice([rng.randint(1, 3000), rng.randint(3000, 200000), size])
              lens += [ln] * rng.choice([1, 1, 2, 5])
          mblobs = [splitmix64_bytes(rng.randint(1, 1 << 30), max(ln, 1)).tobytes()[:ln] for ln in lens]
          mseeds = [rng.randint(0, 1 << 40) for _ in lens]
          want = []
          try:
              for b, s_ in zip(mblobs, mseeds):
                  want.append(O.commit_and_generate_proof(b, s_, ocfg))
          except Exception:
              want = None  # a blob the reference panics on for this configuration: the whole call must report it
          if want is None:
              try:
                  mc.prove_many(mblobs, mseeds, cfg)
                  raise AssertionError(("oracle panicked, prove_many did not", lens, B, last))
              except frieda_amd.FriedaPanic:
                  pass
          else:
              got = mc.prove_many(mblobs, mseeds, cfg)
              for (rr, pp), (orr, opp) in zip(got, want):
                  assert rr == orr and pp.serialize() == opp.serialize(), ("prove_many", lens, B, last, nq, pow_bits)
              assert mc.commit_many(mblobs, B) == [orr for orr, _ in want], ("commit_many", lens, B)
              n_batch += 1
      if size <= 20000 and rng.random() < 0.3:
          cnt = rng.randint(2, 9)
          blobs = [splitmix64_bytes(rng.randint(1, 1 << 30), max(size, 1)).tobytes()[:size] for _ in range(cnt)]
          seeds = [rng.randint(0, 1 << 40) for _ in range(cnt)]
          try:
              got = ctx.commit_and_generate_proof_batch(blobs, seeds, cfg)
          except frieda_amd.FriedaError as e:
              if "device channel" in str(e):
                  continue
              raise
          for b, s_, (rr, pp) in zip(blobs, seeds, got):
              orr, opp = O.commit_and_generate_proof(b, s_, ocfg)
              assert rr == orr and pp.serialize() == opp.serialize(), ("batch", size, B, last, nq, pow_bits)
          n_batch += 1
  if mc is not None:
      mc.close()
  return n_single, n_batch, time.time() - t0, n_big


if __name__ == "__main__":
    a, b, dt, nb = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print(f"fuzz ok: {a} proofs ({nb} of them on 0.4 - 8 MB blobs), {b} batches in {dt:.0f} s")

