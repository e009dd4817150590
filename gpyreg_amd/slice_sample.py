"""Bounded coordinate-wise slice sampler with adaptive widths.

The sampler the reference uses to draw hyperparameter samples (gpyreg/slice_sample.py,
a port of MATLAB ``slicesamplebnd``; Neal 2003, "shrinkage" procedure).  It is a strictly
sequential Markov chain -- every evaluation depends on the previous one -- so it is NOT
batched ("replicas only"); each ``log_f`` call is one device NLL evaluation.

Speculative shrinkage (``options["log_f_batch"]``).  Within one coordinate update the sequence of shrinkage
proposals does not depend on the target's values: a rejected proposal shrinks the interval to itself on its side of
the current point, and the next proposal is one uniform draw in the new interval.  Only WHERE the sequence stops
does.  So the next ``speculate`` proposals are generated ahead, evaluated as ONE device batch (a batch of eight costs
about what a single evaluation does at the sizes the sampler runs at, and its rows carry the bits of single
evaluations), and the first accepted one is taken; the global RNG is then rewound and advanced by exactly the
draws the sequential procedure would have made.  The chain, the widths and the RNG stream afterwards are those of
the sequential sampler; a fit makes about a third of the device calls.

This implementation draws from the global NumPy RNG in the same order as the reference
(one coordinate shuffle per sweep; per coordinate a slice level, an interval offset, then
one uniform per shrinkage proposal), so that with the same seed and the same target values
it walks the same chain.  Kept: bounds, optional stepping-out, width adaptation during
burn-in (x1.2 when the first proposal is accepted, /1.1 after more than three shrinks, then
5 x the burn-in standard deviation), thinning.  Dropped: Metropolis mixing, convergence
diagnostics and the logging front-end, none of which ``GP.fit`` uses.
"""

from __future__ import annotations

import numpy as np


class SliceSampler:
    def __init__(self, log_f, x0, widths=None, LB=None, UB=None, options=None):
        self.log_f = log_f
        self.x0 = np.array(x0, dtype=float).copy()
        if self.x0.ndim > 1:
            raise ValueError("The initial point x0 needs to be a scalar or a 1D array")
        D = self.x0.size
        self.LB = np.full(D, -np.inf) if LB is None else np.array(LB, dtype=float).ravel() * np.ones(D)
        self.UB = np.full(D, np.inf) if UB is None else np.array(UB, dtype=float).ravel() * np.ones(D)
        if not np.all(self.UB >= self.LB):
            raise ValueError("All upper bounds UB need to be equal or greater than lower bounds LB.")
        if np.any(self.x0 < self.LB) or np.any(self.x0 > self.UB):
            raise ValueError("The initial starting point X0 is outside the bounds.")
        span = self.UB - self.LB
        # the first interval may reach one ulp beyond the box (like the reference); the
        # target is -inf there, so such proposals are simply shrunk away
        self.LB_out = np.nextafter(self.LB, -np.inf)
        self.UB_out = np.nextafter(self.UB, np.inf)
        self.base_widths = None if widths is None else np.array(widths, dtype=float).ravel() * np.ones(D)
        if self.base_widths is None:
            w = span / 2
            w[~np.isfinite(w)] = 10.0
        else:
            w = self.base_widths.copy()
        w[self.LB == self.UB] = 1.0
        if np.any(w <= 0) or np.any(~np.isfinite(w)):
            raise ValueError("The widths vector needs to be all positive real numbers.")
        self.widths = w
        options = options or {}
        self.step_out = options.get("step_out", False)
        self.adaptive = options.get("adaptive", True)
        self.log_f_batch = options.get("log_f_batch")  # rows of points -> one value per row
        self.speculate = int(options.get("speculate", 4))
        self.func_count = 0
        self.device_calls = 0

    def _logp(self, x):
        if np.any(x < self.LB) or np.any(x > self.UB):
            return -np.inf, None
        f = self.log_f(x)
        self.func_count += 1
        self.device_calls += 1
        if np.any(np.isnan(f)):
            return -np.inf, f
        return float(np.sum(f)), f

    def _shrink_speculative(self, xx, x_l, x_r, dd, log_u):
        """The shrinkage loop of one coordinate with its proposals evaluated ``speculate`` at a time.
        Returns (accepted coordinate value, log_Px, f_val, number of proposals made)."""
        shrink = 0
        lo, hi = x_l[dd], x_r[dd]
        while True:
            state = np.random.get_state()
            props, ends = [], []
            l, r = lo, hi
            for _ in range(self.speculate):
                xp = np.random.rand() * (r - l) + l
                props.append(xp)
                if xp > xx[dd]:
                    r = xp
                elif xp < xx[dd]:
                    l = xp
                ends.append((l, r))
                if xp == xx[dd]:
                    break  # shrunk onto the current point: the sequence ends here whatever the value
            pts = np.repeat(xx[None, :], len(props), axis=0)
            pts[:, dd] = props
            inside = ~(np.any(pts < self.LB, axis=1) | np.any(pts > self.UB, axis=1))
            vals = np.full(len(props), -np.inf)
            raw = [None] * len(props)
            f = None
            if np.any(inside):
                try:
                    # an array of values, or a callable row -> value (the host-side part of the target, e.g. the
                    # log prior, is then only evaluated for the rows the sequential procedure would have reached)
                    f = self.log_f_batch(pts[inside])
                except np.linalg.LinAlgError:
                    # a proposal the sequential procedure might never have reached is numerically infeasible (its
                    # matrix stays non-positive-definite): hand the rest of this coordinate to the sequential loop.
                    # Anything else -- a HIP error out of the C ABI, a ShardError, a bug in a user kernel -- propagates
                    np.random.set_state(state)
                    x_l[dd], x_r[dd] = lo, hi
                    return None, None, None, shrink
                self.device_calls += 1
            row_of = np.cumsum(inside) - 1
            taken = len(props)
            for k, xp in enumerate(props):
                if inside[k]:
                    fk = f(int(row_of[k])) if callable(f) else f[int(row_of[k])]
                    raw[k] = fk
                    vals[k] = -np.inf if np.isnan(fk) else float(fk)
                if vals[k] > log_u or xp == xx[dd]:
                    taken = k + 1
                    break
            # the RNG stream of the sequential procedure: one draw per proposal actually made
            np.random.set_state(state)
            for _ in range(taken):
                np.random.rand()
            self.func_count += int(np.sum(inside[:taken]))
            shrink += taken
            k = taken - 1
            if vals[k] > log_u or props[k] == xx[dd]:
                # the interval as the sequential loop leaves it (the accepted proposal does not shrink it): later
                # coordinates' step-out evaluations of this sweep read these ends (sample(): x_l, x_r are per sweep)
                x_l[dd], x_r[dd] = ends[k - 1] if k > 0 else (lo, hi)
                return props[k], vals[k], raw[k], shrink
            lo, hi = ends[k]

    def sample(self, N: int, thin: int = 1, burn: int = None):
        xx = self.x0
        D = xx.size
        if burn is None:
            burn = 0 if self.func_count > 0 else round(N / 3)
        if not np.isscalar(thin) or thin <= 0:
            raise ValueError("The thinning factor option needs to be a positive integer.")
        if not np.isscalar(burn) or burn < 0:
            raise ValueError("The burn-in samples option needs to be a non-negative integer.")
        total = N + (N - 1) * (thin - 1)
        samples = np.zeros((N, D))
        f_vals = np.zeros((N, 1))
        sum1, sum2 = np.zeros(D), np.zeros(D)
        log_Px, f_val = self._logp(xx)
        if not np.isfinite(log_Px):
            raise ValueError("The initial starting point X0 needs to evaluate to a real number (not Inf or NaN).")
        perm = np.arange(D)
        for it in range(total + burn):
            x_l, x_r, xprime = xx.copy(), xx.copy(), xx.copy()
            np.random.shuffle(perm)
            for dd in perm:
                if self.LB[dd] == self.UB[dd]:
                    continue
                log_u = log_Px + np.log(np.random.rand())  # slice level
                rr = np.random.rand()  # position of the current point inside the first interval
                x_l[dd] = np.fmax(x_l[dd] - rr * self.widths[dd], self.LB_out[dd])
                x_r[dd] = np.fmin(x_r[dd] + (1 - rr) * self.widths[dd], self.UB_out[dd])
                if self.step_out and self.widths[dd] > 0:
                    # (a width of exactly 0 -- the end-of-burn-in estimate from ONE stored sweep, burn = 2 or 3 -- would
                    # step out by nothing for ever: the reference's loop, :412-417, does not return there)
                    while self._logp(x_l)[0] > log_u:
                        x_l[dd] -= self.widths[dd]
                    while self._logp(x_r)[0] > log_u:
                        x_r[dd] += self.widths[dd]
                shrink = 0
                sequential = self.log_f_batch is None or self.speculate <= 1
                if not sequential:
                    xp, lp, fv, shrink = self._shrink_speculative(xx, x_l, x_r, dd, log_u)
                    if xp is None:
                        sequential = True  # (x_l, x_r, the RNG and the count are where the batches left them)
                    else:
                        xprime[dd], log_Px, f_val = xp, lp, fv
                while sequential:
                    shrink += 1
                    xprime[dd] = np.random.rand() * (x_r[dd] - x_l[dd]) + x_l[dd]
                    log_Px, f_val = self._logp(xprime)
                    if log_Px > log_u:
                        break
                    if xprime[dd] > xx[dd]:
                        x_r[dd] = xprime[dd]
                    elif xprime[dd] < xx[dd]:
                        x_l[dd] = xprime[dd]
                    else:
                        break  # shrunk onto the current point
                if it < burn and self.adaptive:
                    delta = self.UB[dd] - self.LB[dd]
                    if shrink > 3:
                        floor = np.abs(np.spacing(delta)) if np.isfinite(delta) else np.spacing(1)
                        self.widths[dd] = np.maximum(self.widths[dd] / 1.1, floor)
                    elif shrink < 2:
                        self.widths[dd] = np.minimum(self.widths[dd] * 1.2, delta)
                xx[dd] = xprime[dd]
                # (x_l, x_r keep this coordinate's final interval until the next sweep, as in the reference, :385-387:
                # with step_out the later coordinates' interval ends are evaluated THERE, not at the updated point)
            if it >= burn and (it - burn) % thin == 0:
                k = (it - burn) // thin
                samples[k] = xx
                f_vals[k] = f_val
            if burn / 2 <= it < burn:
                sum1 += xx
                sum2 += xx**2
                if it == burn - 1 and self.adaptive:
                    n = np.floor(burn / 2)
                    new_w = np.fmin(5 * np.sqrt(np.maximum(sum2 / n - (sum1 / n) ** 2, 0)),
                                    self.UB_out - self.LB_out)
                    if self.base_widths is None:
                        self.widths = new_w
                    else:
                        self.widths = np.maximum(new_w, np.sqrt(new_w * self.base_widths))
        return {"samples": samples, "f_vals": f_vals, "exit_flag": 0, "log_priors": np.zeros(N),
                "R": None, "eff_N": None}
