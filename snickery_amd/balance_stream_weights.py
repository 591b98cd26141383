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
import copy
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


def balance_stream_weights(synth, n_tune=10, n_valid=10, max_epochs=1000, patience=5, thresh=0.001,
                           eta=0.1, amplifier=1.2, attenuator=0.5, dmax=50.0, dmin=0.000001,
                           weight_floor=0.0, report=print):
    """Runs the balancing loop on `synth` (a snickery_amd.synthesiser.Synthesiser, or any object
    with its interface) and returns a dict: best_weights, join_stream_weights,
    target_stream_weights, losses, contribs, weight_history, validation_mean_scores."""
    assert synth.config['join_cost_weight'] == 1.0
    synth.mode_of_operation = 'stream_weight_balancing'
    synth.verbose = False
    njoin = len(synth.stream_list_join)
    ntarget = len(synth.stream_list_target)

    weights = np.ones(njoin + ntarget)          # initially, unweighted streams
    best_weights = copy.copy(weights)
    best_score = previous_score = float('inf')
    epochs_without_improvement = 0
    lrates = np.ones(weights.shape) * eta
    prev_directions = np.ones(weights.shape)
    losses, contribs, history = [], [], []

    flist = synth.get_sentence_set('tune')
    tune_flist = flist[:n_tune] if n_tune < len(flist) else flist
    valid_flist = flist[n_tune:n_tune + n_valid]
    flist = tune_flist

    def contributions(names):
        cache = _synth_all(synth, names)
        jscores = np.vstack([j for (t, j) in cache])
        tscores = np.vstack([t for (t, j) in cache])
        return mean_nonzero_contributions(jscores, tscores)

    goals = None
    i = -1
    for i in range(max_epochs):
        synth.set_join_weights(weights[:njoin])
        synth.set_target_weights(weights[njoin:])
        if synth.config.get('greedy_search', False):
            synth.get_tree_for_greedy_search()
        mean_scores = contributions(flist)
        if i == 0:
            # target and join sides contribute equally, streams equally within a side (:115-120)
            goal_join = (mean_scores.sum() / 2.0) / njoin
            goal_target = (mean_scores.sum() / 2.0) / ntarget
            goals = np.array([goal_join] * njoin + [goal_target] * ntarget)
        errors = mean_scores - goals
        loss = np.abs(errors).sum()
        report('')
        report('=== iteration %s | loss %s ===' % (i + 1, loss))
        losses.append(loss)
        if loss < previous_score:
            epochs_without_improvement = 0
        else:
            epochs_without_improvement += 1
        if loss < best_score:
            best_score = loss
            best_weights = copy.copy(weights)
        if epochs_without_improvement == patience:
            report('\n   ----> converged (or diverged and ran out of patience)\n')
            break
        if loss < thresh:
            report('\n   ----> loss approaching 0: stop here\n')
            break
        directions = np.sign(-1.0 * errors)       # change weights in the opposite direction of errors
        direction_change = directions * prev_directions
        lrates[direction_change > 0] *= amplifier
        lrates[direction_change < 0] *= attenuator
        lrates = np.clip(lrates, dmin, dmax)
        prev_directions = copy.copy(directions)
        update = directions * lrates
        report('')
        report('     Streams: ' + ' '.join([item.ljust(8) for item in synth.stream_list_join + synth.stream_list_target]))
        report('Prev weights: ' + ' '.join(['%f' % (val) for val in weights.tolist()]))
        weights += update
        weights = np.maximum(weights, weight_floor)
        previous_score = loss
        report('mean contrib: ' + ' '.join(['%f' % (val) for val in mean_scores.tolist()]))
        report('       goals: ' + ' '.join(['%f' % (val) for val in goals.tolist()]))
        report('      errors: ' + ' '.join(['%f' % (val) for val in errors.tolist()]))
        report('      update: ' + ' '.join(['%f' % (val) for val in update.tolist()]))
        report('     weights: ' + ' '.join(['%f' % (val) for val in weights.tolist()]))
        contribs.append(mean_scores)
        history.append(weights.copy())
    if i == max_epochs - 1:
        report('\n   ----> max epochs reached: stop here\n')

    report('')
    report('# ============================================')
    if valid_flist:
        report('# validate found weights on %s sentences (%s ... %s)' % (len(valid_flist), valid_flist[0], valid_flist[-1]))
    synth.set_join_weights(best_weights[:njoin])
    synth.set_target_weights(best_weights[njoin:])
    if synth.config.get('greedy_search', False):
        synth.get_tree_for_greedy_search()
    # the reference scores the TUNE list here as well (balance_stream_weights.py:187)
    validation = contributions(flist)
    report('# mean contribution (validation): ' + ' '.join(['%f' % (val) for val in validation.tolist()]))
    report('')
    report('## Weights found to best balance stream contributions -- you can copy these to config.')
    if flist:
        report('## Weights were found using %s utterances (%s ... %s)' % (len(flist), flist[0], flist[-1]))
    report('join_stream_weights = %s' % (best_weights.tolist()[:njoin]))
    report('target_stream_weights = %s' % (best_weights.tolist()[njoin:]))
    return {'best_weights': best_weights, 'join_stream_weights': best_weights.tolist()[:njoin],
            'target_stream_weights': best_weights.tolist()[njoin:], 'losses': losses,
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
