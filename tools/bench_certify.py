"""`mrr10_match`: the certificate BASELINE.md section 3 asks for inside the driver-run bench line -- the CPU path and the
HIP path on THE SAME inputs, list against list and MRR@10 against MRR@10 (metric code: MEVI/evaluate.py:7-24,
MEVI/ensemble_marco.py:221-240).

  * dense_certificate: a fixed slice of the bench's own data (first 64 queries x first 500 000 corpus rows, which hold
    the planted neighbours) through oracle.dense.ip_topk_exact (CPU, sequential fmaf chains) and through the HIP search;
  * chain_certificate: 32 queries of the C4 chain on a 500 000-row sub-corpus -- tower, dense top-k, NCI beam search,
    fine stage, ensemble -- CPU (oracle/, the host dict path of the ensemble pinned to golden G6) against GPU.

Only bench.py calls this (oracle/ as the checker, never as the thing measured)."""
import time

import numpy as np
import torch


def mrr_at(lists, gts, k=10):
    """MEVI/evaluate.py:7-24: 1 / (first rank of any gt document + 1) when that rank is < k, averaged over queries."""
    tot = 0.0
    for ranked, gt in zip(lists, gts):
        ranked = list(ranked)
        hits = [ranked.index(g) for g in gt if g in ranked]
        if hits and min(hits) < k:
            tot += 1.0 / (min(hits) + 1)
    return tot / max(len(gts), 1)


def dense_certificate(query, docs, planted, topk, nq_s=64, nd_s=500_000):
    from mevi_amd import dense
    from oracle import dense as odense

    nq_s, nd_s = min(nq_s, query.shape[0]), min(nd_s, docs.shape[0])
    q, d = query[:nq_s].contiguous(), docs[:nd_s].contiguous()
    k = min(topk, nd_s)
    gs, gi = dense.DenseIndex(d).search(q, k)
    torch.cuda.synchronize()
    t = time.perf_counter()
    cs, ci = odense.ip_topk_exact(q.cpu().numpy(), d.cpu().numpy(), k)
    cpu_s = time.perf_counter() - t
    gi_h, gs_h = gi.cpu().numpy(), gs.cpu().numpy()
    gts = [[int(p)] for p in planted[:nq_s]]
    m_cpu, m_gpu = mrr_at(ci.tolist(), gts), mrr_at(gi_h.tolist(), gts)
    return {"slice": f"first {nq_s} queries x first {nd_s} corpus rows of the bench's own data (planted ids included), top-{k}",
            "cpu": "oracle.dense.ip_topk_exact (sequential f32 fmaf chains, (score desc, id asc))", "cpu_seconds": round(cpu_s, 2),
            "lists_identical": bool(np.array_equal(ci, gi_h) and np.array_equal(cs.view(np.uint32), gs_h.view(np.uint32))),
            "mrr10_cpu": m_cpu, "mrr10_gpu": m_gpu, "abs_diff": abs(m_cpu - m_gpu)}


def chain_certificate(model, tower, cpu_w, cfgs, docs, codes_h, codebook, ids, mask, planted, M, K, R, topk,
                      n_q=8, nd_s=500_000, oracle_generate=None):
    """cpu_w = (NCI state dict, tower state dict) on the CPU, cfgs = (nci cfg, tower cfg) for oracle.t5.
    `codes_h` i32 [N, M]: the corpus' RQ codes (the offline index artefact both paths read; re-checked on a sample
    against oracle.rq).  `oracle_generate`: (decoded, scores) of oracle.t5.nci_generate for the first n_q queries when
    bench.py has them already (it is the slow part: ~2 s per query on host cores)."""
    from mevi_amd import consumers, dense, fine, metrics, nci, rq
    from oracle import dense as odense
    from oracle import rq as orq
    from oracle import t5 as ot5

    n_q, nd_s = min(n_q, ids.shape[0]), min(nd_s, docs.shape[0])
    dev = docs.device
    i_s, m_s = ids[:n_q], mask[:n_q]
    sub = docs[:nd_s].contiguous()
    sub_codes = np.ascontiguousarray(codes_h[:nd_s])
    k = min(topk, nd_s)
    t0 = time.perf_counter()
    # ---- GPU ----------------------------------------------------------------------------------------------------------------
    qe = tower.encode_query({"input_ids": i_s, "attention_mask": m_s})
    gs, gi = dense.DenseIndex(sub).search(qe, k)
    dec, gsc, _, _ = model.generate(i_s, m_s, num_beams=R)
    bc = nci.decode_token(dec, K).view(n_q, R, M).cpu().numpy()
    index = rq.ClusterIndex.from_codes(sub_codes, K)
    ranked, _ = fine.FineStage(sub, index).rerank(qe, bc)
    gi_h, gs_h = gi.cpu().numpy(), gs.cpu().numpy()
    # ---- CPU ----------------------------------------------------------------------------------------------------------------
    sub_h = sub.cpu().numpy()
    with torch.no_grad():
        ce = ot5.tower_encode(cpu_w[1], cfgs[1], i_s.cpu(), m_s.cpu())
        if oracle_generate is None:
            odec, osc, _ = ot5.nci_generate(cpu_w[0], cfgs[0], i_s.cpu(), m_s.cpu(), R)
        else:
            odec, osc = oracle_generate
    cs, ci = odense.ip_topk_exact(ce.numpy(), sub_h, k)
    obc = ot5.decode_token(odec[:n_q * R], K).view(n_q, R, M).numpy()
    n_chk = min(nd_s, 20_000)
    rq_same = bool(np.array_equal(orq.rq_encode(sub_h[:n_chk], codebook.cpu().numpy()), sub_codes[:n_chk]))
    cluster, mapping = orq.cluster_dict(sub_codes)
    cfine = [odense.fine_stage(ce[i].numpy(), sub_h, cluster, obc[i]) for i in range(n_q)]
    # ---- ensemble + MRR@10 on both sides (alpha .6 beta .03 gamma .02, MEVI/marco_ensemble.sh) ----------------------------------
    qs = [f"q{i}" for i in range(n_q)]
    import chain_c4

    gts = [chain_c4.synthetic_gt(i, planted[i], cfine[i][0]) for i in range(n_q)]      # the chain's own gt rule

    def ensemble(dense_i, dense_s, fine_lists, beams, alpha=0.6):
        dp = {q: dense_i[i].tolist() for i, q in enumerate(qs)}
        ds_ = {q: [float(x) for x in dense_s[i]] for i, q in enumerate(qs)}
        cranks, ncl = metrics.cluster_ranks(dp, {q: beams[i].tolist() for i, q in enumerate(qs)}, metrics.ArrayMapping(sub_codes))
        return [metrics.ensemble_scores(dp[q], ds_[q], cranks[q], [int(x) for x in fine_lists[i][0]],
                                        [float(x) for x in fine_lists[i][1]], ncl, alpha, 0.03, 0.02) for i, q in enumerate(qs)]

    ens_c2 = ens_g2 = None
    try:
        ens_c = ensemble(ci, cs, cfine, obc)
        ens_g = ensemble(gi_h, gs_h, ranked, bc)
        ens_c2 = ensemble(ci, cs, cfine, obc, chain_c4.ALPHA_STRONG)       # where the beam clusters re-order the top 10
        ens_g2 = ensemble(gi_h, gs_h, ranked, bc, chain_c4.ALPHA_STRONG)
        # the device consumers on the GPU lists must agree with the host dict path on the same lists
        fseg = np.concatenate([[0], np.cumsum([len(r[0]) for r in ranked])]).astype(np.int64)
        inp = consumers.EnsembleInputs(qs, torch.arange(n_q + 1, device=dev) * k, gi.reshape(-1), gs.reshape(-1).double(), bc,
                                       sub_codes, (qs, np.arange(n_q, dtype=np.int64), fseg,
                                                   np.concatenate([np.asarray(r[0], np.int64) for r in ranked]),
                                                   np.concatenate([np.asarray(r[1], np.float64) for r in ranked])))
        od, on = inp.ensemble(inp.ranks(), 0.6, 0.03, 0.02)
        od, on, oseg = od.cpu().numpy(), on.cpu().numpy(), inp.out_seg.cpu().numpy()
        dev_same = all(od[oseg[i]:oseg[i] + on[i]].tolist() == ens_g[i] for i in range(n_q))
    except AssertionError as e:      # queries disagreeing on the number of distinct beam clusters (ensemble_marco.py:185-187)
        ens_c = ens_g = None
        dev_same = f"ensemble refused as the reference does: {e}"

    def both(c_lists, g_lists):
        a, b = mrr_at(c_lists, gts), mrr_at(g_lists, gts)
        return {"mrr10_cpu": a, "mrr10_gpu": b, "abs_diff": abs(a - b),
                "top10_identical": float(np.mean([list(c)[:10] == list(g)[:10] for c, g in zip(c_lists, g_lists)]))}

    out = {
        "slice": f"first {n_q} queries of the chain x first {nd_s} corpus rows (planted neighbours included), beams {R}, "
                 f"RQ ({M},{K}), top-{k}",
        "cpu": "oracle.t5 tower + NCI generate (torch fp32), oracle.dense exact chains + fine_stage, host dict ensemble",
        "seconds": round(time.perf_counter() - t0, 1),
        "tower_max_abs_diff": float((qe.cpu() - ce).abs().max()),
        "beams_identical": bool(np.array_equal(bc, obc)),
        "queries": n_q, "beams": int(n_q * R),
        "beams_differing_from_oracle": int((bc != obc).any(-1).sum()),     # near-tie swaps (0 when beams_identical)
        "beam_score_max_abs_diff": float(np.abs(np.asarray(gsc[:n_q * R]) - osc[:n_q * R].numpy()).max()),
        "rq_codes_identical_on_sample": rq_same,
        "dense": both(ci.tolist(), gi_h.tolist()),
        "fine": both([c[0].tolist() for c in cfine], [np.asarray(r[0]).tolist() for r in ranked]),
        "device_ensemble_equals_host_dict_path": dev_same,
        "note": "tower embeddings differ by f32 summation order (tower_max_abs_diff), so documents inside near-ties may swap: "
                "the north star's bar is MRR@10 within 1e-4",
    }
    if ens_c is not None:
        out["ensemble"] = both(ens_c, ens_g)
    if ens_c2 is not None:
        out["ensemble_alpha%g" % chain_c4.ALPHA_STRONG] = both(ens_c2, ens_g2)
    out["mrr10_within_1e-4"] = bool(all(v["abs_diff"] <= 1e-4 for k_, v in out.items() if isinstance(v, dict) and "abs_diff" in v))
    return out
