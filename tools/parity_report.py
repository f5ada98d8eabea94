#!/usr/bin/env python
"""Per-stage parity of the HIP path against the golden fixtures (GPU box).  Prints max-rel errors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from labelanything_amd.models import Lam
from labelanything_amd.episodes import make_episode
from tests.cases import CASES
from labelanything_amd.engine import PRECISE_DEFAULT, PRECISE_FULL, PRECISE_WIDE, PRECISE_WIDE_PLANES
from tests.helpers import load_golden, rel_err, reference_logits, argmax_disagreement, pct_rel_err


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    variants = [(torch.float16, torch.float32, PRECISE_DEFAULT), (torch.float16, torch.float32, ()),
                (torch.float16, torch.float32, ("patch", "qkv", "proj", "lin1", "lin2", "neck")),
                (torch.float16, None, PRECISE_DEFAULT), (torch.bfloat16, torch.float32, PRECISE_DEFAULT)]
    if "--quick" in sys.argv:
        variants = variants[:2]
    if "--decoder" in sys.argv:     # exact-fp32 MFMA decoder against the fp16 plane-pair (3-product) image side
        variants = [(torch.float16, torch.float32, "auto"), (torch.float16, "f16x2", "auto")]
    if "--groups" in sys.argv:      # the default against leaner group sets (which planes are worth their MFMA passes)
        variants = [(torch.float16, torch.float32, g) for g in (PRECISE_FULL, PRECISE_WIDE, PRECISE_WIDE_PLANES, ("patch", "neck"),
                                                                ("patch", "vmean", "neck"), ("patch", "qkv", "neck"))]
    for dt, ddt, precise in variants:
        for name, case in CASES.items():
            if only and name not in only:
                continue
            gold, meta = load_golden(name)
            lam = Lam(case["cfg"], seed=case["weight_seed"], compute_dtype=dt, decoder_dtype=ddt, precise=precise).cuda()
            lam.selected_rows = gold.get("selected_rows")
            batch = make_episode(**case["episode"])
            t0 = time.time()
            try:
                seg, pe = lam._forward(batch)
                out = lam.forward_argmax(batch)
                torch.cuda.synchronize()
            except Exception as e:
                print(f"[{name} {dt}] FAILED: {type(e).__name__}: {e}")
                import traceback; traceback.print_exc()
                continue
            t1 = time.time() - t0
            d = lam.cfg.embed_dim
            errs = {}
            eng = lam.engine()
            if "query_embedding" in gold or "query_embedding_sample" in gold:
                e32, b, n, g = lam._embeddings_nhwc(batch, True)
                q = e32.view(b, n, g * g, d)[:, 0].permute(0, 2, 1).reshape(b, d, g, g)
                if "query_embedding" in gold:
                    errs["query_emb"] = rel_err(q, gold["query_embedding"])
                else:
                    errs["query_emb"] = rel_err(q[:, ::8, ::4, ::4], gold["query_embedding_sample"])
            errs["class_emb"] = rel_err(pe["class_embeddings"], gold["class_embeddings"])
            errs["cls_ex_emb"] = rel_err(pe["class_examples_embeddings"], gold["class_examples_embeddings"])
            errs["low_res"] = rel_err(seg, gold["low_res_logits"])
            if "logits" in gold:
                errs["logits"] = rel_err(out["logits"], gold["logits"])
            am = out["argmax"].cpu()
            ref_am = gold["argmax"].long()
            mism = int((am != ref_am).sum())
            errs["argmax_mismatch_frac"] = mism / ref_am.numel()
            am2 = out["logits"].argmax(1).cpu()
            errs["fused_argmax_vs_torch"] = int((am2 != am).sum())
            ref_logits = reference_logits(case, gold, batch)
            errs["logits_vs_ref"] = rel_err(out["logits"], ref_logits)
            errs["logits_p99.9_elementwise"] = pct_rel_err(out["logits"], ref_logits)
            n_diff, n_real = argmax_disagreement(out["logits"], ref_am, ref_logits, margin_rel=2e-3)
            errs["argmax_diff_outside_2e-3_margin"] = n_real
            ptag = "auto=" + "+".join(lam.precise) if precise == "auto" else ("none" if not precise else "+".join(precise))
            print(f"[{name} enc={str(dt)[6:]} dec={(ddt if isinstance(ddt, str) else str(ddt)[6:]) if ddt else 'same'} precise={ptag}] {t1:.2f}s  " + "  ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in errs.items()), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
