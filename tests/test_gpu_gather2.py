"""The row gather of SURVEY §8e with MORE THAN ONE rank, on the one GPU a test box has.  RCCL refuses two ranks on one
device, so for these tests liblgmi is pointed (LGMI_RCCL_LIB) at tests/helpers/fake_rccl.cpp — the same ten entry
points carried between processes through files — and everything above it is the product: lgmi_comm_gather and its two
halves, the same-batch reductions, site bases, the socket rendezvous.  Rank 0 compares what it gathered with unsharded
runs, bit for bit (tests/helpers/gather2_worker.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def fake_rccl(tmp_path_factory):
    out = str(tmp_path_factory.mktemp('fake_rccl') / 'librccl_fake.so')
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    subprocess.run(['g++', '-O1', '-shared', '-fPIC', '-D__HIP_PLATFORM_AMD__', '-I%s/include' % rocm,
                    os.path.join(HERE, 'helpers', 'fake_rccl.cpp'), '-o', out, '-L%s/lib' % rocm, '-lamdhip64',
                    '-Wl,-rpath,%s/lib' % rocm], check=True)
    return out


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('world', [2, 3])
def test_gather_between_ranks_sharing_one_gpu(fake_rccl, world):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                   LGMI_RDZV_PORT=str(port), LGMI_RCCL_LIB=fake_rccl)
        env.pop('TORCHELASTIC_USE_AGENT_STORE', None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'helpers', 'gather2_worker.py')], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()                                       # exactly the processes started here
        # the stand-in's mailbox lives in /dev/shm (memory): nothing of it may outlive the test, passed or not; its
        # directory name carries the pid of the rank that made the unique id (rank 0)
        import glob
        import shutil
        for d in glob.glob('/dev/shm/frccl_%d_*' % procs[0].pid):
            shutil.rmtree(d, ignore_errors=True)
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (rank, se[-3000:])
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line['ok'] and line['world'] == world and len(line['checked']) == 5
