"""K3 (optimal Huffman tables on the device) against the oracle's jpeg_gen_optimal_table restatement, fed with
synthetic symbol statistics instead of an image's: few and many live symbols (the merge loop runs on 1, 2, 3 or 5
registers per lane depending on how many there are), ties (the larger-symbol rule), counts on both sides of the
2^22 single-key limit, and distributions skewed enough to need the K.3 code-length limiting."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _device_tables(mij, torch, hist4):
    """Builds the four tables from `hist4` (4 x 257 uint32) on the device: transform a tiny image (so that the encoder has
    coefficients), overwrite the statistics, run the entropy stage (K3 first), read the tables back."""
    W = H = 16
    img = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    d_hist = torch.zeros(4 * 257, dtype=torch.int32, device="cuda")
    with mij.Encoder(W, H, 90, True, 0) as enc:
        enc.set_histogram_buffer(d_hist.data_ptr())
        enc.transform(img.data_ptr(), W * 3, "rgb", 0, 0)
        torch.cuda.synchronize()
        d_hist.copy_(torch.from_numpy(hist4.astype(np.uint32).view(np.int32).reshape(-1)).cuda())
        torch.cuda.synchronize()
        enc.entropy(0)
        enc.result()
        return enc.debug_tables()


def _check(mij, oracle, hist4):
    import torch
    got = _device_tables(mij, torch, hist4)
    for t in range(4):
        bits, vals = oracle.gen_optimal_table(hist4[t, :256])
        assert got[t, :17].tolist() == [int(b) for b in bits], ("bits", t)
        n = len(vals)
        assert got[t, 17:17 + n].tolist() == [int(v) for v in vals], ("vals", t)


def _random_hist(rng, live, lo, hi, dc_rows=True):
    h = np.zeros((4, 257), np.uint64)
    for t in range(4):
        nsym = 12 if (dc_rows and t in (0, 2)) else 256      # DC tables: sizes 0..11 only
        k = min(live, nsym)
        idx = rng.choice(nsym, size=k, replace=False)
        h[t, idx] = rng.integers(lo, hi, size=k, dtype=np.uint64)
    return h.astype(np.uint32)


@pytest.mark.parametrize("live", [1, 2, 3, 12, 40, 63, 64, 65, 127, 128, 129, 162, 191, 192, 193, 256])
def test_tables_match_oracle_by_live_symbols(mij, oracle, live):
    rng = np.random.default_rng(live)
    # (libjpeg's procedure only ever picks nodes of at most 10^9, so a table's counts must sum to less than that)
    for lo, hi in ((1, 50), (1, 3_000_000), (4_000_000 // live + 1, 4_400_000 // live + 3), (1, 900_000_000 // live)):
        _check(mij, oracle, _random_hist(rng, live, lo, hi))


def test_tables_ties_and_skew(mij, oracle):
    h = np.zeros((4, 257), np.uint32)
    h[:, :200] = 7                                   # every count equal: only the tie rule decides
    h[0, 12:] = 0; h[2, 12:] = 0
    _check(mij, oracle, h)
    h = np.zeros((4, 257), np.uint32)
    fib = [1, 1]
    while len(fib) < 40:
        fib.append(fib[-1] + fib[-2])                # Fibonacci counts: the deepest possible tree, lengths far beyond 16
    for t in (1, 3):
        h[t, 5:45] = np.array(fib, np.uint64).astype(np.uint32)
    h[0, :12] = [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144]
    h[2, :12] = 3
    _check(mij, oracle, h)
    h = np.zeros((4, 257), np.uint32)
    h[:, 0] = 1                                      # a single live symbol per table
    _check(mij, oracle, h)
