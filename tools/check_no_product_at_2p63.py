#!/usr/bin/env python3
"""check_no_product_at_2p63.py: the encode kernels' shortcut route (|t| < 2^51, encode_device.hpp) never meets |fl(r * 10^f)| == 2^63.
On that route r is an integer with |r| <= 2^51; the exact product r * 10^f (10^f exact in double for f <= 22) rounds to 2^63 only from
[2^63 - 512, 2^63 + 1024] (half an ulp below, half an ulp above with the tie going to the even mantissa of 2^63).  Enumerate, for every f the
tables hold, the integers r whose product lies there: none has |r| <= 2^51.  (Negative r: the mirror image.)  Float: |r| <= 2^22 against
[2^31 - 64, 2^31 + 128] (the float kernels test the int32 product itself; listed for completeness.)"""
bad = []
for f in range(0, 23):
    lo, hi = -(-(2**63 - 512) // 10**f), (2**63 + 1024) // 10**f
    bad += [(f, r) for r in range(lo, min(hi, 2**51) + 1)]
    print(f"f = {f:2d}: r in [{lo}, {hi}] -> {'none' if lo > hi else ('all above 2^51' if lo > 2**51 else 'SOME <= 2^51')}")
print("shortcut route can meet |prod| == 2^63:", bool(bad))
raise SystemExit(1 if bad else 0)
