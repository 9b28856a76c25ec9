"""The C++ host mirror (fhe-study_amd/host/arith.hpp) restates the reference's own
NTT-path tests over the C ABI.  The program is built on CPU (it needs only
include/fhe_ntt.h and the .so) and run on the GPU."""
import os
import subprocess

import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "fhe-study_amd", "host")


def _build(pkg, exe="test_arith"):
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, exe)


def test_host_mirror_builds_against_the_c_abi(pkg):
    exe = _build(pkg)
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_host_mirror_reference_tests(pkg):
    exe = _build(pkg)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host C++ tests passed" in r.stdout


def test_gfhe_mirror_builds_against_the_c_abi(pkg):
    assert os.path.exists(_build(pkg, "test_gfhe"))


@pytest.mark.gpu
def test_gfhe_mirror_reference_tests(pkg):
    """gfhe.hpp: TR / GLWE / GLev / key_switch over the host-buffer N3 entry points; restates
    gfhe/src/glwe.rs:582-626 (test_key_switch) and checks every surface against its definition"""
    exe = _build(pkg, "test_gfhe")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host C++ gfhe tests passed" in r.stdout


@pytest.mark.gpu
def test_plain_c_example_runs(pkg, tmp_path):
    """examples/rq_mul.c: the boundary from C99 — the reference's product KAT and a 1000-polynomial
    round trip through the host-buffer entry points"""
    exe = str(tmp_path / "rq_mul")
    libdir = os.path.join(ROOT, "fhe-study_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-o", exe, os.path.join(ROOT, "examples", "rq_mul.c"),
                           "-L" + libdir, "-lfhe_ntt", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit-exact" in r.stdout


def test_bfv_mirror_builds_against_the_c_abi(pkg):
    assert os.path.exists(_build(pkg, "test_bfv"))


@pytest.mark.gpu
def test_bfv_mirror_reference_tests(pkg):
    """bfv.hpp: RLWE::tensor / RLWE::mul / tmp_naive_mul over the C ABI; restates
    bfv/src/lib.rs:504-601 (test_tensor, test_mul_relin: decrypt(c1*c2) == m1*m2)"""
    exe = _build(pkg, "test_bfv")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host C++ bfv tests passed" in r.stdout


def test_tfhe_mirror_builds_against_the_c_abi(pkg):
    assert os.path.exists(_build(pkg, "test_tfhe"))


@pytest.mark.gpu
def test_tfhe_mirror_reference_tests(pkg):
    """tfhe.hpp: Tn * Tn and TGGSW * TGLWE over the C ABI; restates tfhe/src/tggsw.rs:157-196
    (test_external_product: decode(decrypt(TGGSW(m1) * TGLWE(m2))) == m1*m2)"""
    exe = _build(pkg, "test_tfhe")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all host C++ tfhe tests passed" in r.stdout


def test_persistent_transform_protocols_simulated(pkg):
    """csrc/persist_sched.hpp (the ticket arithmetic the kernels compile) driven by CPU models of the two persistent
    kernels' workgroup state machines under a random scheduler: no deadlock from ONE workgroup per queue upwards, every
    part run exactly once, ring slots never rewritten before they were read (fhe-study_amd/host/test_persist_sched.cpp)."""
    exe = _build(pkg, "test_persist_sched")
    r = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all persist schedule tests passed" in r.stdout


def test_c_abi_without_a_device_or_with_one(pkg):
    """every path of include/fhe_ntt.h that needs no device — plan construction and cache (16 threads), the reference's
    argument checks in the reference's order, shard arithmetic, switches — and the compute entry points, which must return
    FHE_E_NO_DEVICE here and the right words on the GPU box (fhe-study_amd/host/test_nodevice.cpp; `make san` runs the same
    program under ASan + UBSan against a host-only build: profiles/r04_host_sanitizers.txt)"""
    exe = _build(pkg, "test_nodevice")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all no-device tests passed" in r.stdout
