"""Times image_prepare_resnet on the device against the CPU oracle for photo-sized inputs.
Run under `rocprofv3 --kernel-trace --stats` to get the two kernels' durations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.clip import image_prepare_resnet
from oracle.binding import load_oracle, orc_image_prepare_resnet

orc = load_oracle()
for (h, w) in ((1080, 1920), (3000, 4000), (480, 640)):
    img = synth.photo_u8(1, h, w)
    image_prepare_resnet(img)  # warm
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        out = image_prepare_resnet(img)
    t_gpu = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    ref = orc_image_prepare_resnet(orc, img)
    t_cpu = time.perf_counter() - t0
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    print(f"{h}x{w}: device path (H2D + 2 kernels + D2H, synchronous) {t_gpu*1e3:.3f} ms/image = {h*w*3/t_gpu/1e9:.2f} GB/s of source bytes; "
          f"CPU oracle (1 thread) {t_cpu*1e3:.1f} ms/image; bit-identical")
