#!/usr/bin/env python3
"""Stream-weight balancing on the MI355X search path.

Python-3 counterpart of the reference's ``script/balance_stream_weights.py`` (the whole script,
:1-226): starting from unit weights, synthesise a tune set in 'stream_weight_balancing' mode, measure
each stream's mean non-zero contribution to the selected paths (:94-111), and move every weight
against its error with a sign-based, per-weight step size that grows while the error keeps its
sign and shrinks when it flips (:143-163), until the loss stops improving (:126-141).

The search throughput is what makes the loop practical (the reference loops ``synth_utt`` up to
1000 times over the tune set, balance_stream_weights.py:82-92): re-weighting is one O(N*D) device
pass (no index rebuild) and a Viterbi configuration hands the whole tune set to
``snk_knn_viterbi_batch`` (Synthesiser.synth_utts_bulk).

    python -m snickery_amd.balance_stream_weights -c voice.cfg
"""
from argparse import ArgumentParser

import numpy as np


def mean_nonzero_contributions(join_scores, target_scores):
    """balance_stream_weights.py:94-111: per column, the mean over the strictly positive entries
    (0.0 for a column without any); join columns first."""
    means = []
    for scores in (join_scores, target_scores):
        for column in range(scores.shape[1]):
            vals = scores[:, column]
            vals = vals[vals > 0.0]
            means.append(vals.sum() / vals.shape[0] if vals.shape[0] else 0.0)
    return np.array(means)


class SignStepState(object):
    """Per-weight sign-step rule of the reference's tuning loop (balance_stream_weights.py:143-163): every
    weight moves against the sign of its error by its own step size, which is multiplied by `grow` while
    the sign repeats, by `shrink` when it flips, and kept inside [step_min, step_max].  One vectorised
    update per iteration; the object also keeps the best weights seen and the stall counter that ends the
    search (:126-141)."""

    def __init__(self, n, step0, grow, shrink, step_min, step_max, floor):
        self.w = np.ones(n)                      # unweighted streams to begin with (:66)
        self.step = np.full(n, float(step0))
        self.last_sign = np.ones(n)
        self.grow, self.shrink, self.lo, self.hi, self.floor = grow, shrink, step_min, step_max, floor
        self.best_w, self.best_loss, self.prev_loss, self.stalled = self.w.copy(), np.inf, np.inf, 0

    def observe(self, loss):
        """Book-keeping of one measured loss; returns the number of iterations without improvement."""
        self.stalled = 0 if loss < self.prev_loss else self.stalled + 1
        if loss < self.best_loss:
            self.best_loss, self.best_w = loss, self.w.copy()
        return self.stalled

    def advance(self, errors, loss):
        """Moves the weights one step against `errors`; returns the applied update."""
        sign = np.sign(-errors)
        agreement = sign * self.last_sign
        self.step = np.clip(self.step * np.where(agreement > 0, self.grow, np.where(agreement < 0, self.shrink, 1.0)),
                            self.lo, self.hi)
        self.last_sign = sign
        update = sign * self.step
        self.w = np.maximum(self.w + update, self.floor)
        self.prev_loss = loss
        return update


class _Report(object):
    """The reference script's console lines (same labels and number formats), through one callable."""

    def __init__(self, emit, stream_names):
        self.emit, self.names = emit, stream_names

    @staticmethod
    def _row(values):
        return ' '.join('%f' % v for v in np.asarray(values).tolist())

    def iteration(self, number, loss):
        self.emit('')
        self.emit('=== iteration %s | loss %s ===' % (number, loss))

    def stop(self, why):
        self.emit('\n   ----> %s\n' % why)

    def step(self, before, contrib, goals, errors, update, after):
        self.emit('')
        self.emit('     Streams: ' + ' '.join(name.ljust(8) for name in self.names))
        self.emit('Prev weights: ' + self._row(before))
        for label, values in (('mean contrib', contrib), ('       goals', goals), ('      errors', errors),
                              ('      update', update), ('     weights', after)):
            self.emit('%s: %s' % (label, self._row(values)))

    def summary(self, valid_names, tune_names, validation, njoin, best):
        self.emit('')
        self.emit('# ============================================')
        if valid_names:
            self.emit('# validate found weights on %s sentences (%s ... %s)' % (len(valid_names), valid_names[0], valid_names[-1]))
        self.emit('# mean contribution (validation): ' + self._row(validation))
        self.emit('')
        self.emit('## Weights found to best balance stream contributions -- you can copy these to config.')
        if tune_names:
            self.emit('## Weights were found using %s utterances (%s ... %s)' % (len(tune_names), tune_names[0], tune_names[-1]))
        self.emit('join_stream_weights = %s' % (best.tolist()[:njoin]))
        self.emit('target_stream_weights = %s' % (best.tolist()[njoin:]))


def _apply(synth, weights, njoin):
    synth.set_join_weights(weights[:njoin])
    synth.set_target_weights(weights[njoin:])
    if synth.config.get('greedy_search', False):
        synth.get_tree_for_greedy_search()


def _measure(synth, names):
    """Mean non-zero contribution of every stream over the utterances `names`."""
    scored = _synth_all(synth, names)
    return mean_nonzero_contributions(np.vstack([j for (_t, j) in scored]), np.vstack([t for (t, _j) in scored]))


def balance_stream_weights(synth, n_tune=10, n_valid=10, max_epochs=1000, patience=5, thresh=0.001,
                           eta=0.1, amplifier=1.2, attenuator=0.5, dmax=50.0, dmin=0.000001,
                           weight_floor=0.0, report=print):
    """Runs the balancing loop on `synth` (a snickery_amd.synthesiser.Synthesiser, or any object
    with its interface) and returns a dict: best_weights, join_stream_weights,
    target_stream_weights, losses, contribs, weight_history, validation_mean_scores."""
    assert synth.config['join_cost_weight'] == 1.0
    synth.mode_of_operation = 'stream_weight_balancing'
    synth.verbose = False
    njoin, ntarget = len(synth.stream_list_join), len(synth.stream_list_target)
    out = _Report(report, synth.stream_list_join + synth.stream_list_target)
    state = SignStepState(njoin + ntarget, eta, amplifier, attenuator, dmin, dmax, weight_floor)

    names = synth.get_sentence_set('tune')
    tune = names[:n_tune] if n_tune < len(names) else names
    valid = names[n_tune:n_tune + n_valid]

    goals, losses, contribs, history = None, [], [], []
    epoch = -1
    for epoch in range(max_epochs):
        _apply(synth, state.w, njoin)
        contrib = _measure(synth, tune)
        if goals is None:
            # both sides are to contribute the same, and every stream of a side the same (:115-120)
            half = contrib.sum() / 2.0
            goals = np.concatenate([np.full(njoin, half / njoin), np.full(ntarget, half / ntarget)])
        errors = contrib - goals
        loss = np.abs(errors).sum()
        out.iteration(epoch + 1, loss)
        losses.append(loss)
        if state.observe(loss) == patience:
            out.stop('converged (or diverged and ran out of patience)')
            break
        if loss < thresh:
            out.stop('loss approaching 0: stop here')
            break
        before = state.w.copy()
        update = state.advance(errors, loss)
        out.step(before, contrib, goals, errors, update, state.w)
        contribs.append(contrib)
        history.append(state.w.copy())
    if epoch == max_epochs - 1:                    # also after a stop on the last epoch, as the reference (:174-175)
        out.stop('max epochs reached: stop here')

    best = state.best_w
    _apply(synth, best, njoin)
    validation = _measure(synth, tune)          # the reference scores the TUNE list here too (:187)
    out.summary(valid, tune, validation, njoin, best)
    return {'best_weights': best, 'join_stream_weights': best.tolist()[:njoin],
            'target_stream_weights': best.tolist()[njoin:], 'losses': losses,
            'contribs': np.vstack(contribs) if contribs else np.zeros((0, njoin + ntarget)),
            'weight_history': history, 'validation_mean_scores': validation}


def _synth_all(synth, names):
    """[(tscores, jscores)] of the utterances: one batched device pass when the front end offers it."""
    if hasattr(synth, 'synth_utts_bulk'):
        return synth.synth_utts_bulk(names, synth_type='tune')
    return [synth.synth_utt(name, synth_type='tune') for name in names]


def main(argv=None):
    a = ArgumentParser()
    a.add_argument('-c', dest='config_fname', required=True)
    a.add_argument('--device', type=int, default=0)
    opts = a.parse_args(argv)
    from .synthesiser import Synthesiser
    synth = Synthesiser(opts.config_fname, device=opts.device)
    try:
        balance_stream_weights(synth)
    finally:
        synth.close()


if __name__ == '__main__':
    main()
