#!/usr/bin/env python3
"""Where a bounded window should keep its ONE backward state (window.cpp place_rows): total descent, in part-lengths,
of the backward passes of a window of P parts processed from its start, for the midpoint rule of round 3 and for
anchors c * sqrt(parts left) above the current part."""
import math


def total(P, rule):
    A, cost, from_end = None, 0, 0
    for p in range(1, P + 1):
        if A is not None and A >= p:
            cost += A - p
        else:
            cost += P - p
            from_end += 1
            a = rule(p, P)
            A = a if p < a < P else None
    return cost, from_end


for P in (24, 31, 37, 45):
    print(P, "midpoint", total(P, lambda p, P: p + (P - p) // 2),
          {c: total(P, lambda p, P, c=c: p + max(1, int(round(c * math.sqrt(P - p))))) for c in (1.0, 1.2, 1.414, 1.7, 2.0)})
