"""The step-time model of tools/predict_scaling.py, importable without a GPU: one rank's measured costs + an iteration
profile + the assumed link costs -> its phase times.  tools/predict_workloads.py applies it to other workloads' iteration
profiles from the ranks measured for the Taylor-Green run (the per-iteration costs do not depend on the flow)."""

ASSUMED = {
    "what": "link-side costs no one-GPU run can measure; everything else in this file is measured",
    "xgmi_link_GBps_per_direction": 76.8,  # 153.6 GB/s bidirectional per link (7 links per GPU, point to point)
    "link_efficiency": 0.7,
    "exchange_latency_us": 12.0,           # grouped ncclSend/ncclRecv between two GPUs over what the self-loop already pays
    "allreduce_extra_latency_us": {"1": 0.0, "2": 8.0, "4": 12.0, "8": 17.0},  # small-message ncclAllReduce, over the 1-rank launch
}


# the xGMI-window transport (OX_TRANSPORT=p2p): the self-loop already pays the push and pull kernels and the one-kernel
# all-reduce; what a real link adds is the one-way latency of uncached remote stores + flag (no library call, no launch)
ASSUMED_P2P = {
    "what": "link-side costs no one-GPU run can measure, for the xGMI-window transport; everything else is measured",
    "xgmi_link_GBps_per_direction": 76.8,
    "link_efficiency": 0.7,
    "exchange_latency_us": 4.0,
    "allreduce_extra_latency_us": {"1": 0.0, "2": 3.0, "4": 4.0, "8": 5.0},
}


def predict(m, profile, P, ASSUMED=ASSUMED):
    """Step time (ms) of one rank from its measured costs, the iteration profile and the modelled link costs."""
    bw = ASSUMED["xgmi_link_GBps_per_direction"] * ASSUMED["link_efficiency"] * 1e9
    lat = ASSUMED["exchange_latency_us"] * 1e-3 if P > 1 else 0.0
    ar = ASSUMED["allreduce_extra_latency_us"][str(P)] * 1e-3 if P > 1 else 0.0

    def xch(nvals, ncomp):  # ms the link adds to one halo exchange (the slower of this rank's send and receive sides)
        return 0.0 if P == 1 else lat + 1e3 * 8.0 * ncomp * nvals / bw
    xu3 = xch(max(m.get("send_max_u", 0), m.get("recv_max_u", 0)), 3)
    xu1 = xch(max(m.get("send_max_u", 0), m.get("recv_max_u", 0)), 1)
    xp1 = xch(max(m.get("send_max_p", 0), m.get("recv_max_p", 0)), 1)
    # exchanges / all-reduces per iteration of the partitioned defaults: merged BiCGStab 2 mat-vecs + 2 points,
    # single-reduction or merged CG 1 mat-vec + 1 point
    ph = {}
    ph["assemble_first"] = m["assemble_first_ms"]
    ph["velocity_tentative_assemble"] = m["tentative_assemble_ms"] + (xp1 if P > 1 else 0.0) * 0  # (ps ghosts are current)
    ph["velocity_tentative_solve"] = (m["bcgs3"]["fixed_ms"] + profile["tent3"] * (m["bcgs3"]["iter_ms"] + 2 * xu3 + 2 * ar)
                                      + profile["tent1"] * (m["bcgs1"]["iter_ms"] + 2 * xu1 + 2 * ar)
                                      + (m["bcgs1"]["fixed_ms"] if profile["tent1"] > 0 else 0.0) + xu3)
    ph["pressure_assemble"] = m["pressure_assemble_ms"]
    ph["pressure_solve"] = m["cgP1"]["fixed_ms"] + profile["pressure"] * (m["cgP1"]["iter_ms"] + xp1 + ar) + xp1 + 2 * ar
    ph["velocity_update"] = (m["update_rhs_ms"] + m["cgM3"]["fixed_ms"] + profile["upd3"] * (m["cgM3"]["iter_ms"] + xu3 + ar)
                             + profile["upd1"] * (m["cgM1"]["iter_ms"] + xu1 + ar)
                             + (m["cgM1"]["fixed_ms"] if profile["upd1"] > 0 else 0.0) + xu3)
    ph["step_vector_work"] = m["step_vector_work_ms"]
    return ph
