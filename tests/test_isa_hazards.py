"""CPU: the generated gfx950 ISA has no TRANS result read by the very next VALU instruction.  hipcc guards that hazard for its
own instructions, not for an inline-asm reader; round 2 met one (profiles/r02_insitu_costs.md) -- the results stayed bit-exact
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
