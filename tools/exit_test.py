import sys
sys.path.insert(0, '.')
mode = sys.argv[1]
from dnncancerannotator_amd import _lib, device
if mode == 'load':
    _lib.load()
elif mode == 'init':
    device.init_device(0)
elif mode == 'count':
    print(device.device_count())
elif mode == 'model':
    device.init_device(0)
    m = device.DeviceModel('unet', 1, 32, 32, 2, 3, 3, padding='same')
    m.close()
print('done', mode, flush=True)
