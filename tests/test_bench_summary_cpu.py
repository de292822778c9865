"""The compact `summary` object of the bench line (benchlib/headline.py summary_of): last key, under 600 bytes with full-size
values, None where an extra did not run, result_ok_all false as soon as one check fails.  CPU only."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from benchlib.headline import second_metric, summary_of  # noqa: E402


def full_line():
    return {"ms_per_step": 1.0370123456, "result_ok": True, "alu_roofline": {"frac_vs_raw_mad_step": 0.5912345},
            "extra": {"C2_msm_2e16": {"ms_per_msm_one_at_a_time": 0.27512345, "ms_per_msm_two_in_flight": 0.1781234, "ms_per_msm_three_in_flight": 0.14212345, "result_ok": True},
                      "C3_ipa_prover": {"value": 0.0231234567, "deterministic": True, "with_fixed_generators": {"seconds": 0.0220123456}},
                      "C4_aggregated_range_proof": {"prove_s": 0.00687, "verify_s": 0.00512, "verified": True, "wrong_commitment_rejected": True},
                      "C5_batch_verify": {"value": 13912345.678, "batch_latency_s": 0.0021234, "accepted": True, "corrupted_batch_rejected": True, "batches_in_flight": 8,
                                          "link": {"GBps": 35.512345, "peak_GBps": 63.0}, "wire_format_2": {"value": 16912345.6},
                                          "wire_format_3": {"value": 19912345.6, "accepted": True, "corrupted_batch_rejected": True},
                                          "batch_prover": {"proves_per_s": 508123.4, "byte_identical_to_single_proof_prover_on_sample": True,
                                                           "aggregated": {"byte_identical_to_AggregNIRangeProver_on_sample": True}}}}}


def test_summary_is_compact_and_complete():
    out = full_line()
    sm = summary_of(out)
    assert len(json.dumps(sm)) < 600
    assert sm["result_ok_all"] is True and sm["checks"] == 11
    assert sm["C2_ms_two"] == 0.1781 and sm["C3_s"] == 0.02312 and sm["C5_one_batch_ms"] == 2.123 and sm["C5_link_GBps"] == 35.51
    assert all(v is not None for v in sm.values())
    m2 = second_metric(out)
    assert m2["value2"] == 13912345.678 and m2["unit2"] == "verifies/s" and "8 batches in flight" in m2["metric2"]


def test_summary_without_extras_and_with_a_failed_check():
    sm = summary_of({"ms_per_step": 1.0, "result_ok": True})
    assert sm["C3_s"] is None and sm["C5_verifies_per_s"] is None and sm["result_ok_all"] is True and sm["checks"] == 1
    assert second_metric({"ms_per_step": 1.0}) == {}
    out = full_line()
    out["extra"]["C4_aggregated_range_proof"]["verified"] = False
    assert summary_of(out)["result_ok_all"] is False
    out = full_line()
    out["extra"]["C3_ipa_prover"] = {"error": "RuntimeError: x"}
    sm = summary_of(out)
    assert sm["C3_s"] is None and sm["result_ok_all"] is True and sm["checks"] == 10
