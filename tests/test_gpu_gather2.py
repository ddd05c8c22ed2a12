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
                   LGMI_RDZV_PORT=str(port), LGMI_RCCL_LIB=fake_rccl, LGMI_ALLOW_RCCL_STANDIN='1')
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
    assert line['comm_info']['stand_in'] is True and line['comm_info']['nranks'] == world


def test_stand_in_is_refused_without_the_explicit_switch(fake_rccl):
    """LGMI_RCCL_LIB alone (a stale environment variable) must not put the product's gather on a stand-in"""
    code = ("import sys; sys.path.insert(0, %r); import lgmi\n"
            "e = lgmi.Engine(0)\n"
            "try:\n"
            "    e.comm_unique_id(); print('ACCEPTED')\n"
            "except lgmi._lib.LgmiError as x:\n"
            "    print('REFUSED' if 'LGMI_ALLOW_RCCL_STANDIN' in str(x) else 'OTHER: %%s' %% x)\n"
            % os.path.join(os.path.dirname(HERE), 'l-giremi_amd'))
    env = dict(os.environ, LGMI_RCCL_LIB=fake_rccl)
    env.pop('LGMI_ALLOW_RCCL_STANDIN', None)
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.stdout.strip().splitlines()[-1] == 'REFUSED', (r.stdout, r.stderr[-2000:])
