"""SURVEY.md section 8c rule for fp32 OMP parity: "picks identical unless top-2 score gap < 1e-4 * max".  TEST INFRASTRUCTURE.

`omp_pick_audit` replays OMP_estimate.m:7-23 in float64 ALONG THE DEVICE'S OWN PICKS: at every iteration it forms the residual of
the picks made so far (pinv refit, :17-18), scores all atoms (|S' r|, :14) and requires the device's next pick to be the arg-max
or -- counted and reported -- within 1e-4 * max of it.  A pick that is neither fails the test, however few there are.  Returns
(near_ties, H) where H = fft of the refit on the device's picks: what the estimate must then equal to fp32 accuracy."""
import numpy as np


def omp_pick_audit(oracle, Y, S, picks_1based, Nfft, gap=1e-4):
    Y = np.asarray(Y, dtype=np.complex128).ravel()
    S = np.asarray(S, dtype=np.complex128)
    idx, near, r = [], 0, Y.copy()
    x = np.zeros(0, dtype=np.complex128)
    for it, k1 in enumerate(picks_1based):
        k = int(k1) - 1
        sc = np.abs(S.conj().T @ r)
        best = int(np.argmax(sc))
        if k != best:
            assert sc[best] - sc[k] < gap * sc[best], (
                f"iteration {it}: device picked atom {k + 1} (score {sc[k]:.6g}) but atom {best + 1} scores {sc[best]:.6g}: "
                f"gap {(sc[best] - sc[k]) / sc[best]:.3g} of the maximum, not a near-tie")
            near += 1
        idx.append(k)
        A = S[:, idx]
        x = oracle._pinv_matlab(A) @ Y
        r = Y - A @ x
    h = np.zeros(int(Nfft), dtype=np.complex128)
    for k, v in zip(idx, x):
        h[k] = v                                         # a later duplicate overwrites (OMP_estimate.m:31-33)
    return near, np.fft.fft(h)
