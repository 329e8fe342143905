"""Independent pin of the box-overlap core.

The reference's IoU / GIoU (src/dprt/utils/iou.py:72-118, 121-210) call the absent third-party
``pytorch3d.ops.box3d_overlap``; the oracle (oracle/dprt_oracle.py::giou3d_yaw, oracle/metric_oracle.py) and the
HIP kernel (dpft_giou3d_yaw_f32) both use a BEV polygon clip x z-overlap.  This file checks that geometry against a
second implementation that shares NO code and no formulation with it: the general 3-D one pytorch3d uses -- each
box is the intersection of the six half-spaces of its faces (normals taken from the corner coordinates by cross
products, no yaw parametrisation), the overlap is scipy's ``HalfspaceIntersection`` of the twelve, its volume
``ConvexHull.volume``; feasibility / the interior point come from a Chebyshev-centre linear programme.
"""
import numpy as np
import pytest
import torch
from scipy.optimize import linprog
from scipy.spatial import ConvexHull, HalfspaceIntersection

from oracle import dprt_oracle as O
from oracle import metric_oracle as MO

# faces of the unit box by corner index, corner k = bits (x = k&1, y = k>>1&1, z = k>>2&1): NOT the reference's order
_FACES = ((0, 2, 4), (1, 5, 3), (0, 4, 1), (2, 3, 6), (0, 1, 2), (4, 6, 5))


def _corners(c, s, yaw):
    """(8, 3) corners from centre / size / yaw through a rotation MATRIX applied to the bit-indexed unit cube."""
    bits = np.array([[(k >> 0) & 1, (k >> 1) & 1, (k >> 2) & 1] for k in range(8)], dtype=np.float64) - 0.5
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0.0], [np.sin(yaw), np.cos(yaw), 0.0], [0.0, 0.0, 1.0]])
    return (bits * np.asarray(s, dtype=np.float64)) @ R.T + np.asarray(c, dtype=np.float64)


def _halfspaces(P):
    """[n | -n.p] rows (n.x + d <= 0 inside) of the six faces, outward normals from cross products of corner edges."""
    ctr = P.mean(0)
    hs = []
    for a, b, c in _FACES:
        n = np.cross(P[b] - P[a], P[c] - P[a])
        n = n / np.linalg.norm(n)
        if np.dot(n, ctr - P[a]) > 0:
            n = -n
        hs.append(np.concatenate((n, [-np.dot(n, P[a])])))
    return np.array(hs)


def _overlap_volume(P1, P2):
    hs = np.concatenate((_halfspaces(P1), _halfspaces(P2)), 0)
    # Chebyshev centre: max r s.t. n.x + r <= -d
    res = linprog(c=[0, 0, 0, -1.0], A_ub=np.concatenate((hs[:, :3], np.ones((12, 1))), 1), b_ub=-hs[:, 3],
                  bounds=[(None, None)] * 3 + [(0, None)], method="highs")
    if res.status != 0 or res.x[3] <= 1e-11:
        return 0.0
    verts = HalfspaceIntersection(hs, res.x[:3]).intersections
    return float(ConvexHull(verts).volume)


def _iou_giou_independent(c1, s1, a1, c2, s2, a2):
    P1, P2 = _corners(c1, s1, a1), _corners(c2, s2, a2)
    v1, v2 = float(np.prod(s1)), float(np.prod(s2))
    vol = _overlap_volume(P1, P2)
    allc = np.concatenate((P1, P2), 0)
    evol = float(np.prod(allc.max(0) - allc.min(0)))
    iou = vol / (v1 + v2 - vol) if vol > 0 else 0.0
    uni = (v1 + v2 - vol) if iou != 0 else 0.0            # iou.py:187-188: uni = vol / iou, 0 where iou == 0
    return iou, iou - (evol - uni) / evol


def box_cases(seed=7, n_random=2000):
    """(c1, s1, a1, c2, s2, a2) rows; shared with tests/test_gpu_kernels.py (the GPU twin of this test)."""
    rng = np.random.default_rng(seed)
    rows = []
    for _ in range(n_random):
        c1 = rng.normal(size=3) * 2.0
        s1 = rng.uniform(0.5, 5.0, size=3)
        c2 = c1 + rng.normal(size=3) * rng.choice([0.3, 1.5, 4.0])
        s2 = rng.uniform(0.5, 5.0, size=3)
        rows.append((c1, s1, rng.uniform(-np.pi, np.pi), c2, s2, rng.uniform(-np.pi, np.pi)))
    one = np.array([2.0, 3.0, 1.5])
    z3 = np.zeros(3)
    for yaw in (0.0, 0.3, np.pi / 2, -2.0):
        rows.append((z3, one, yaw, z3, one, yaw))                                   # identical boxes
        rows.append((z3, one, yaw, z3, one * 0.4, yaw + 0.7))                       # containment, rotated inner box
        rows.append((z3, one * 0.25, yaw, np.array([0.1, -0.1, 0.05]), one, yaw))   # contained the other way round
    for ax in range(3):                                                             # touching faces / edges, disjoint
        sh = np.zeros(3); sh[ax] = one[ax]
        rows.append((z3, one, 0.0, sh, one, 0.0))
        rows.append((z3, one, 0.0, sh * 1.5, one, 0.0))
        rows.append((z3, one, 0.0, sh * 0.5, one, 0.0))                             # half overlap along one axis
    rows.append((z3, one, 0.0, np.array([2.0, 3.0, 0.0]), one, 0.0))                # touching along a vertical edge
    rows.append((z3, np.array([2.0, 2.0, 2.0]), 0.0, z3, np.array([2.0, 2.0, 2.0]), np.pi / 4))   # octagon prism
    rows.append((z3, np.array([4.0, 1.0, 1.0]), 0.0, z3, np.array([4.0, 1.0, 1.0]), np.pi / 2))   # crossed bars
    return rows


def _t(rows):
    c1, s1, a1, c2, s2, a2 = zip(*rows)
    f = lambda x: torch.tensor(np.array(x), dtype=torch.float64)
    return f(c1), f(s1), f(a1), f(c2), f(s2), f(a2)


def test_overlap_core_vs_halfspace_intersection_random_and_edge_cases():
    rows = box_cases()
    assert len(rows) >= 2000
    c1, s1, a1, c2, s2, a2 = _t(rows)
    n_overlap = 0
    worst_iou = worst_giou = 0.0
    for k, r in enumerate(rows):
        iou_i, giou_i = _iou_giou_independent(*r)
        g = float(O.giou3d_yaw(c1[k:k + 1], s1[k:k + 1], a1[k:k + 1], c2[k:k + 1], s2[k:k + 1], a2[k:k + 1])[0, 0])
        iou_m, giou_m = MO.iou_giou(c1[k:k + 1], s1[k:k + 1], a1[k:k + 1], c2[k:k + 1], s2[k:k + 1], a2[k:k + 1])
        n_overlap += iou_i > 0
        worst_iou = max(worst_iou, abs(float(iou_m[0, 0]) - iou_i))
        worst_giou = max(worst_giou, abs(g - giou_i), abs(float(giou_m[0, 0]) - giou_i))
        assert abs(float(iou_m[0, 0]) - iou_i) < 1e-9, (k, r, float(iou_m[0, 0]), iou_i)
        assert abs(g - giou_i) < 1e-9 and abs(float(giou_m[0, 0]) - giou_i) < 1e-9, (k, r, g, float(giou_m[0, 0]), giou_i)
    assert n_overlap > 600, n_overlap                # the random set is not mostly disjoint pairs
    print(f"overlapping pairs {n_overlap}/{len(rows)}, worst |d iou| {worst_iou:.2e}, worst |d giou| {worst_giou:.2e}")


def test_overlap_core_known_answers():
    """Hand-computable volumes, through the independent code AND the oracle."""
    z3 = np.zeros(3)
    cube = np.array([2.0, 2.0, 2.0])
    # cube vs itself turned by 45 degrees: regular octagon of inradius 1 (area 8 (sqrt 2 - 1)) x height 2
    v = _overlap_volume(_corners(z3, cube, 0.0), _corners(z3, cube, np.pi / 4))
    assert abs(v - 16.0 * (np.sqrt(2.0) - 1.0)) < 1e-12
    # crossed 4 x 1 x 1 bars: 1 x 1 x 1
    bar = np.array([4.0, 1.0, 1.0])
    assert abs(_overlap_volume(_corners(z3, bar, 0.0), _corners(z3, bar, np.pi / 2)) - 1.0) < 1e-12
    iou, giou = _iou_giou_independent(z3, bar, 0.0, z3, bar, np.pi / 2)
    assert abs(iou - 1.0 / 7.0) < 1e-12 and abs(giou - (1.0 / 7.0 - (16.0 - 7.0) / 16.0)) < 1e-12
    t = lambda x: torch.tensor(np.array([x]), dtype=torch.float64)
    g = float(O.giou3d_yaw(t(z3), t(bar), t(0.0).reshape(1), t(z3), t(bar), t(np.pi / 2).reshape(1))[0, 0])
    assert abs(g - giou) < 1e-12
    # shared face: volume exactly 0 -> the reference's quirk giou = 0 - (evol - 0) / evol = -1
    iou, giou = _iou_giou_independent(z3, cube, 0.0, np.array([2.0, 0, 0]), cube, 0.0)
    assert iou == 0.0 and giou == -1.0


@pytest.mark.parametrize("bad", [[0.0, 2.0, 2.0], [2.0, 0.0, 2.0], [1e-3, 1e-1, 2.0]])
def test_invalid_boxes_keep_the_reference_convention(bad):
    """_check_nonzero (iou.py:39-69): a face-triangle area <= 1e-4 -> iou 0, giou -1 whatever the geometry."""
    t = lambda x: torch.tensor([x], dtype=torch.float64)
    iou, giou = MO.iou_giou(t([0.0, 0, 0]), t(bad), torch.zeros(1, dtype=torch.float64),
                            t([0.0, 0, 0]), t([2.0, 2, 2]), torch.zeros(1, dtype=torch.float64))
    assert float(iou) == 0.0 and float(giou) == -1.0
    assert float(O.giou3d_yaw(t([0.0, 0, 0]), t(bad), torch.zeros(1, dtype=torch.float64),
                              t([0.0, 0, 0]), t([2.0, 2, 2]), torch.zeros(1, dtype=torch.float64))) == -1.0
