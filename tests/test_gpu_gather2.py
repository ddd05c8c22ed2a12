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


# (a test box admits 6 processes on its card, the test runner being one: 4 leaves a margin; 8 ranks: tests/test_dist_gloo.py)
# form: the compact gather (row_i never travels, row_j only from ranks that skipped a pair, 16-bit counts: the default
# between shards of one batch) and the plain one (LGMI_GATHER_LEGACY=1: every array as it is)
@pytest.mark.parametrize('world,form', [(2, 'compact'), (3, 'compact'), (4, 'compact'), (4, 'plain')])
def test_gather_between_ranks_sharing_one_gpu(fake_rccl, world, form):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                   LGMI_RDZV_PORT=str(port), LGMI_RCCL_LIB=fake_rccl, LGMI_ALLOW_RCCL_STANDIN='1')
        env.pop('LGMI_GATHER_LEGACY', None)
        if form == 'plain':
            env['LGMI_GATHER_LEGACY'] = '1'
        env['LGMI_TEST_GATHER_FORM'] = form
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
    assert line['ok'] and line['world'] == world and len(line['checked']) == 7
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


def _bench_two_ranks(extra_env):
    root = os.path.dirname(HERE)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
           '--workload', 'small_dense_2kx20k', '--shuffles', '50']
    env = dict(os.environ, LGMI_BENCH_DEVICE='0', LGMI_COMM_INIT_TIMEOUT='90', LGMI_GATHER_TIMEOUT='200', **extra_env)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_two_ranks_over_the_stand_in_reports_what_rccl_says(fake_rccl):
    """the driver's N > 1 launch of bench.py, two ranks on the one GPU through the stand-in: exit code 0, the gathered
    rows equal the unsharded run, and the line carries the communicator's own account (lgmi_comm_info) — with
    stand_in = true, so that such a line can never pass for a measurement over RCCL"""
    r, line = _bench_two_ranks({'LGMI_RCCL_LIB': fake_rccl, 'LGMI_ALLOW_RCCL_STANDIN': '1'})
    assert r.returncode == 0, r.stderr[-3000:]
    assert line and line['n_gpus'] == 2 and line['value'] and 'degraded' not in line
    assert line['rccl']['stand_in'] is True and line['rccl']['nranks'] == 2 and line['rccl']['initialised']
    ver = line.get('verify') or line.get('gather_after_permutation', {}).get('verify')
    assert ver and ver['equal_to_unsharded'] is True


def test_bench_starts_its_own_ranks_when_no_launcher_did(fake_rccl):
    """`python3 bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment (how a driver that launches the N = 1 bench
    with plain python may well launch N > 1): the process starts its two ranks itself as fresh children, makes no GPU call
    of its own, relays rank 0's one JSON line and exits 0 only if both ranks do — here over the stand-in on the one GPU"""
    root = os.path.dirname(HERE)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
           '--workload', 'small_dense_2kx20k', '--shuffles', '50']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT', 'MASTER_ADDR',
                                                            'TORCHELASTIC_USE_AGENT_STORE', 'TORCHELASTIC_RUN_ID')}
    env.update(LGMI_BENCH_DEVICE='0', LGMI_COMM_INIT_TIMEOUT='90', LGMI_GATHER_TIMEOUT='200', LGMI_RCCL_LIB=fake_rccl,
               LGMI_ALLOW_RCCL_STANDIN='1')
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] and 'degraded' not in line
    assert line['rccl']['stand_in'] is True and line['rccl']['nranks'] == 2
    ver = line.get('verify') or line.get('gather_after_permutation', {}).get('verify')
    assert ver and ver['equal_to_unsharded'] is True
    # a failing rank is a failing bench: the same launch over the real librccl (two ranks on one device are refused)
    env.pop('LGMI_RCCL_LIB')
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


def test_bench_two_ranks_without_a_communicator_is_not_a_success():
    """two ranks on ONE device over the real librccl: the communicator cannot come up (RCCL refuses the duplicate
    device).  The kernel-only figure is still printed, but as `kernel_only_value` of a line marked `degraded` with no
    `value`, and the exit code is not 0 (a driver must not record this as an N-GPU result)"""
    env = {k: v for k, v in {'LGMI_RCCL_LIB': ''}.items()}
    r, line = _bench_two_ranks(env)
    assert r.returncode != 0
    assert line and line['degraded'] == 'rccl_init_failed' and line['value'] is None and line['kernel_only_value'] > 0
    assert 'NOT gathered' in line['config']['gather']


@pytest.mark.parametrize('world', [2, 3])
def test_cli_on_several_ranks_writes_the_reference_tables(fake_rccl, tmp_path, world):
    """lgmi.cli under torch.distributed.run (--gpus N): footprints dealt to the ranks in contiguous runs, every rank
    extracts and computes its own, the pair rows gathered over the communicator onto rank 0 (here the stand-in, all
    ranks on the one GPU; world 3 leaves a rank without footprints) — the files rank 0 writes are the single-process
    ones, i.e. what the reference's footprint_bulk_calculation computes (tests/golden/cli.json)"""
    sys.path.insert(0, HERE)
    from test_cli import compare_cli_outputs, regions_fixture, write_inputs
    root = os.path.dirname(HERE)
    bam, fa, vcf = write_inputs(tmp_path, regions_fixture())
    prefix = str(tmp_path / 'out')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), '-m', 'lgmi.cli', '-b', bam, '-c', 'chrA', 'chrB', '-o', prefix, '--genome_fasta', fa,
           '--snp_bcf', vcf, '--mi_calculation_only', '--skip_strand_correction', '--gpus', str(world), '--device', '0',
           '--n_shuffles', '20', '--seed', '4']
    env = dict(os.environ, LGMI_RCCL_LIB=fake_rccl, LGMI_ALLOW_RCCL_STANDIN='1',
               PYTHONPATH=os.path.join(root, 'l-giremi_amd') + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    mi = compare_cli_outputs(prefix)
    assert list(mi.columns)[7:] == ['p_perm'] and ((mi['p_perm'] > 0) & (mi['p_perm'] <= 1)).all()
    assert not [f for f in os.listdir(tmp_path) if '.rank' in f]          # the per-rank parts are gone
    # the same p-values as one process computes (the Philox streams are keyed by the pair, not by the rank)
    from lgmi import cli
    one = str(tmp_path / 'one')
    cli.main(['-b', bam, '-c', 'chrA', 'chrB', '-o', one, '--genome_fasta', fa, '--snp_bcf', vcf, '--mi_calculation_only',
              '--skip_strand_correction', '--n_shuffles', '20', '--seed', '4'])
    import pandas as pd
    pd.testing.assert_frame_equal(pd.read_table(one + '.mi.txt'), pd.read_table(prefix + '.mi.txt'))


def test_cli_mip_table_on_two_ranks_equals_the_single_process_file(fake_rccl, tmp_path):
    """--mip_table with --gpus 2 (advice r3: the multi-rank gather of the site table — astype(str) and a float64 round trip
    over the socket group — had no test): the mismatch table with mean_mi and mip that rank 0 writes is the file one process
    writes, byte for byte, and so are the three --mi_calculation_only files"""
    import filecmp
    sys.path.insert(0, HERE)
    from test_cli import regions_fixture, write_inputs
    root = os.path.dirname(HERE)
    bam, fa, vcf = write_inputs(tmp_path, regions_fixture())
    two, one = str(tmp_path / 'two'), str(tmp_path / 'one')
    common = ['-b', bam, '-c', 'chrA', 'chrB', '--genome_fasta', fa, '--snp_bcf', vcf, '--mi_calculation_only',
              '--skip_strand_correction', '--mip_table', '--n_shuffles', '20', '--seed', '4']
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), '-m', 'lgmi.cli', '-o', two, '--gpus', '2', '--device', '0'] + common
    env = dict(os.environ, LGMI_RCCL_LIB=fake_rccl, LGMI_ALLOW_RCCL_STANDIN='1',
               PYTHONPATH=os.path.join(root, 'l-giremi_amd') + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    from lgmi import cli
    cli.main(['-o', one] + common)
    for ext in ('.mismatch_mip.txt', '.mi.txt', '.removed.txt', '.strand.txt'):
        assert filecmp.cmp(one + ext, two + ext, shallow=False), ext
