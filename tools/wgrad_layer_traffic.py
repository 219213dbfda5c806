"""GPU-box aid: the grouped weight-gradient launch of the 96^3 step taken apart — one eager step in which every weight's
descriptors are issued as a launch of their OWN (same kernels, same plans), so that a counter pass attributes fetched bytes per layer.
  run:     rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -o p -- python3 tools/wgrad_layer_traffic.py run LAYERS.json
  report:  python3 tools/wgrad_layer_traffic.py report LAYERS.json OUT"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(out):
    import torch
    import bench
    from vae_segmentation_amd import ops
    from vae_segmentation_amd import train as T
    joint, img, lab = bench.build(96, "bf16", 0)
    layers = []
    real_issue = ops._issue_wgrads

    def issue_one_by_one(entries):
        by_dw = collections.OrderedDict()
        for e in entries:
            by_dw.setdefault(e[0].dw, []).append(e)
        keep = []
        for dw, es in by_dw.items():
            d = es[0][0]
            if RECORD[0]:
                layers.append({"n": d.n, "dp": d.dp, "hp": d.hp, "wp": d.wp, "m_ch": d.m_ch, "c_ch": d.c_ch, "m_real": d.m_real, "c_real": d.c_real, "kind": d.kind,
                               "uses": len(es), "alg_bytes": sum(e[2] for e in es), "flops": sum(e[3] for e in es), "co": d.reserved_})
            keep.append(real_issue(es))
        return keep[-1]

    RECORD = [False]
    ops._issue_wgrads = issue_one_by_one
    for it in range(2):
        RECORD[0] = it == 1
        for p in joint.Seg.parameters():
            p.grad = None
        loss, _ = T.joint_train_losses(joint, img, lab, lambda_vae=0.1)
        loss.backward()
        torch.cuda.synchronize()
    json.dump(layers, open(out, "w"), indent=1)


def report(layers_json, d):
    layers = json.load(open(layers_json))
    f = (glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    main = [r for r in rows if r["Kernel_Name"].startswith(("void g3b_uber_kernel", "void g3b_group_kernel"))]
    main = main[-len(layers):]                                    # the recorded (second) step's launches, in issue order
    red = [r for r in rows if r["Kernel_Name"].startswith("g3_reduce_group_kernel")][-len(layers):]
    print("%-34s %5s %9s %9s %6s" % ("layer (n,d,h,w) c->m kind uses", "", "alg MB", "fetch MB", "ratio"))
    ta = tf = 0.0
    for L, r, rr in zip(layers, main, red):
        fb = 2 * 1024 * float(r["Counter_Value"])
        name = "(%d,%d,%d,%d) %d->%d k%d x%d" % (L["n"], L["dp"], L["hp"], L["wp"], L["c_ch"], L["m_ch"], L["kind"], L["uses"])
        print("%-34s %5s %9.2f %9.2f %6.2f   (+ reduction fetch %.2f MB)" % (name, "", L["alg_bytes"] / 1e6, fb / 1e6, fb / max(L["alg_bytes"], 1), 2 * 1024 * float(rr["Counter_Value"]) / 1e6))
        ta += L["alg_bytes"]; tf += fb
    print("total: algorithmic %.1f MB, fetched %.1f MB (%.2f x)" % (ta / 1e6, tf / 1e6, tf / ta))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3])
