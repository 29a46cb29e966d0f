"""Independent NumPy restatement of semi-global matching (test helper, tiny images only).

Written from the algorithm description, vectorised over d with Python loops
over pixels: a second opinion on the C oracle that shares no code with it."""
import numpy as np

MAXC = 32767


def _planes(I, ft):
    I = I.astype(np.int32)
    H, W = I.shape
    Iu = np.vstack([I[:1], I[:-1]])
    Id = np.vstack([I[1:], I[-1:]])
    g = np.full((H, W), ft, np.int32)
    g[:, 1:-1] = np.clip(2 * (I[:, 2:] - I[:, :-2]) + (Iu[:, 2:] - Iu[:, :-2]) + (Id[:, 2:] - Id[:, :-2]), -ft, ft) + ft
    r = I.copy()
    r[:, 0] = ft
    r[:, -1] = ft
    return g, r


def _interval(p):
    l = p.copy()
    l[:, 1:] = (p[:, 1:] + p[:, :-1]) // 2
    r = p.copy()
    r[:, :-1] = (p[:, :-1] + p[:, 1:]) // 2
    return np.minimum(np.minimum(l, r), p), np.maximum(np.maximum(l, r), p)


def cost_volume(L, R, D=64, P2=2400, ft=15, half=2):
    H, W = L.shape
    W1 = W - D
    pix = np.zeros((H, W1, D), np.int32)
    for (p1, p2), shift in zip(zip(_planes(L, ft), _planes(R, ft)), (0, 2)):
        u0, u1 = _interval(p1)
        v0, v1 = _interval(p2)
        u, uu0, uu1 = p1[:, D:], u0[:, D:], u1[:, D:]
        for d in range(D):
            v, vv0, vv1 = p2[:, D - d:W - d], v0[:, D - d:W - d], v1[:, D - d:W - d]
            c0 = np.maximum(0, np.maximum(u - vv1, vv0 - u))
            c1 = np.maximum(0, np.maximum(v - uu1, uu0 - v))
            pix[:, :, d] += np.minimum(c0, c1) >> shift
    pad = np.pad(pix, ((half, half), (half, half), (0, 0)), mode="edge")
    C = np.full((H, W1, D), P2, np.int32)
    for j in range(2 * half + 1):
        for i in range(2 * half + 1):
            C += pad[j:j + H, i:i + W1]
    return C


DIRS5 = [(-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0)]
DIRS8 = DIRS5 + [(1, 1), (0, 1), (-1, 1)]


def path(C, P1, P2, dx, dy):
    H, W1, D = C.shape
    Lr = np.zeros((H, W1, D), np.int64)
    ys = range(H) if dy <= 0 else range(H - 1, -1, -1)
    for y in ys:
        xs = range(W1) if (dx < 0 or (dx == 0)) else range(W1 - 1, -1, -1)
        if dy != 0:
            xs = range(W1)
        for x in xs:
            px, py = x + dx, y + dy
            Lp = Lr[py, px] if (0 <= px < W1 and 0 <= py < H) else np.zeros(D, np.int64)
            delta = Lp.min() + P2
            lo = np.concatenate([[MAXC], Lp[:-1]]) + P1
            hi = np.concatenate([Lp[1:], [MAXC]]) + P1
            Lr[y, x] = C[y, x] + np.minimum(np.minimum(Lp, lo), np.minimum(hi, delta)) - delta
    return Lr


def aggregate(C, P1=600, P2=2400, dirs=DIRS5):
    S = np.zeros(C.shape, np.int64)
    for dx, dy in dirs:
        S += path(C, P1, P2, dx, dy)
    return np.clip(S, -32768, 32767).astype(np.int32)


def wta(S, W, D=64, uniq=10, d12=1):
    H, W1, _ = S.shape
    disp = np.full((H, W), -16, np.int32)
    for y in range(H):
        disp2 = np.full(W, -16, np.int32)
        cost2 = np.full(W, MAXC, np.int32)
        for x in range(W1 - 1, -1, -1):
            s = S[y, x]
            best = int(np.argmin(s))                    # first minimum == lowest d on ties
            ms = int(s[best])
            if ms >= MAXC:
                continue
            far = np.abs(np.arange(D) - best) > 1
            if np.any((s * (100 - uniq) < ms * 100) & far):
                continue
            x2 = x + D - best
            if cost2[x2] > ms:
                cost2[x2] = ms
                disp2[x2] = best
            if 0 < best < D - 1:
                den = max(int(s[best - 1] + s[best + 1] - 2 * ms), 1)
                num = int(s[best - 1] - s[best + 1]) * 16 + den
                q = abs(num) // (2 * den)
                d16 = best * 16 + (q if num >= 0 else -q)      # C division truncates toward zero
            else:
                d16 = best * 16
            disp[y, x + D] = d16
        for x in range(D, W):
            d1 = disp[y, x]
            if d1 == -16:
                continue
            da, db = d1 >> 4, (d1 + 15) >> 4
            xa, xb = x - da, x - db
            if (0 <= xa < W and disp2[xa] >= 0 and abs(disp2[xa] - da) > d12 and
                    0 <= xb < W and disp2[xb] >= 0 and abs(disp2[xb] - db) > d12):
                disp[y, x] = -16
    return disp


def guided(depth_lo, guide, r, eps, bilinear):
    """guided filter via explicit window means (float64)"""
    I = guide.astype(np.float64) / 255.0
    p = bilinear

    def box(a):
        H, W = a.shape
        out = np.empty_like(a)
        for y in range(H):
            for x in range(W):
                out[y, x] = a[max(y - r, 0):y + r + 1, max(x - r, 0):x + r + 1].mean()
        return out

    mI, mp = box(I), box(p)
    a = (box(I * p) - mI * mp) / (box(I * I) - mI * mI + eps)
    b = mp - a * mI
    return box(a) * I + box(b)
