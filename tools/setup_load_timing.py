import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import lambdaworks_kzg_amd as K
for _ in range(4):
    t = time.perf_counter(); ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt'); dt = time.perf_counter() - t
    print("load_trusted_setup_file: %.1f ms" % (dt * 1e3)); ts.free()
