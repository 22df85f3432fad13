"""CPU: the generated gfx950 ISA has no TRANS result read by the very next VALU instruction.  hipcc guards that hazard for its
own instructions, not for an inline-asm reader; round 2 met one (profiles/history/r02_insitu_costs.md) -- the results stayed bit-exact
through the exact fallback, so only this scan (and the exact_reruns statistic on the GPU) can see it."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_trans_result_is_read_by_the_next_valu_instruction():
    csrc = os.path.join(ROOT, "ascendpathtracing_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asm"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "check_hazards.py"), os.path.join(csrc, "render_kernels.s")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_the_scan_sees_readers_with_operand_modifiers_and_across_labels():
    """a last source followed by modifiers (op_sel_hi:[1,0], clamp) and a reader behind a fall-through label were not
    seen by the first form of the scan."""
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    import check_hazards
    hit = ["v_rsq_f32_e32 v1, v2", "v_pk_mul_f32 v[4:5], v[2:3], v[0:1] op_sel_hi:[1,0]"]
    assert len(check_hazards.scan(hit)) == 1
    assert len(check_hazards.scan(["v_rcp_f32_e32 v7, v2", ".LBB0_3:", "v_add_f32_e64 v9, v3, v7 clamp"])) == 1
    assert len(check_hazards.scan(["v_rsq_f32_e32 v1, v2", "s_nop 0", "v_mul_f32_e32 v3, v1, v1"])) == 0
    assert len(check_hazards.scan(["v_rsq_f32_e32 v1, v2", "v_rcp_f32_e32 v3, v1"])) == 0        # TRANS reader: no hazard
    assert len(check_hazards.scan(["v_rsq_f32_e32 v1, v2", "v_mul_f32_e32 v1, v4, v5"])) == 0   # writes v1, does not read it


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_sample_queue_kernel_has_no_static_lds():
    """render_frame_queue8_kernel (pt_queue.h) addresses its ray pool from LDS address 0: its exec-masked refill block uses
    immediate ds_read offsets, which is only right while the kernel has NO static LDS in front of its dynamic region."""
    import re
    csrc = os.path.join(ROOT, "ascendpathtracing_amd", "csrc")
    subprocess.run(["make", "-s", "-C", csrc, "asm"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(os.path.join(csrc, "render_kernels.s")).read()
    found = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*render_frame_queue8_kernel\S*)\n(.*?)\.end_amdhsa_kernel", text, re.S):
        found += 1
        assert re.search(r"\.amdhsa_group_segment_fixed_size 0\b", m.group(2)), m.group(1)
    assert found >= 2
