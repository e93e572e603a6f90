"""The arithmetic behind the split form of the head kernels (popcorn_hip.h: pc_set_head_split; head.hip: hs_split3 / hs_split_pair),
restated on the CPU: an fp32 number is EXACTLY the sum of three bf16 numbers, and the six partial products the kernels keep reproduce
the fp32 product to within one fp32 rounding."""
import torch


def _split3(x):
    a1 = x.to(torch.bfloat16).to(torch.float32)
    r1 = x - a1                                  # exact in fp32
    a2 = r1.to(torch.bfloat16).to(torch.float32)
    r2 = r1 - a2
    a3 = r2.to(torch.bfloat16).to(torch.float32)
    return a1, a2, a3, r2 - a3


def test_three_bf16_numbers_hold_an_fp32_number_exactly():
    g = torch.Generator().manual_seed(7)
    x = torch.cat([torch.randn(200000, generator=g) * 10.0 ** torch.randint(-6, 7, (200000,), generator=g).float(),
                   torch.tensor([0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.1754944e-38, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24])])
    a1, a2, a3, rest = _split3(x)
    assert torch.all(rest == 0)                                                   # nothing is left after the third piece
    assert torch.equal(a1.double() + a2.double() + a3.double(), x.double())        # ... and the pieces sum to x without rounding
    # every piece is a bf16 number (8 mantissa bits): the fp32 <-> bf16 round trip leaves it alone
    for a in (a1, a2, a3):
        assert torch.equal(a.to(torch.bfloat16).to(torch.float32), a)


def test_six_partial_products_reproduce_the_fp32_product():
    g = torch.Generator().manual_seed(8)
    a = torch.randn(400000, generator=g)
    b = torch.randn(400000, generator=g)
    A, B = _split3(a)[:3], _split3(b)[:3]
    exact = a.double() * b.double()
    # the kept terms (pl + q <= 2), each exact in fp32 (8 x 8 mantissa bits), summed here in float64 to isolate the truncation
    kept = sum(A[p].double() * B[q].double() for p in range(3) for q in range(3) if p + q <= 2)
    rel = ((kept - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
    assert rel < 2.0 ** -22, rel                  # dropped: a2 b3 + a3 b2 + a3 b3 <= 3 * 2^-24 of the product
    # an fp32 multiply-add rounds once per product at 2^-24: the same order
    fp32 = (a * b).double()
    rel32 = ((fp32 - exact).abs() / exact.abs().clamp_min(1e-30)).max().item()
    assert rel < 8 * rel32 + 1e-9
