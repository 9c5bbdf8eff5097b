"""Board power / shader clock of ONE HIP device, read from sysfs by a host thread (measurement plumbing for bench.py and
tools/power_probe.py; nothing on the product path imports it).

MI355X boards are capped at 1 400 W; under chip-wide matrix load on real (full-mantissa) data the kernels of this path sit AT that
cap (profiles/r06_power_probe.txt), so a bench line that carries the mean board power of its timed region says how much of the
time the loop was bound by the power manager rather than by a schedule.  The device is matched by PCI address
(`/sys/class/drm/card*/device` -> 0000:bb:dd.f): other cards of the node belong to other tenants.
"""
import glob
import os
import threading
import time


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ''


def pci_address(device_index=0):
    """'dddd:bb:dd.f' of a HIP device, or None."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return '%04x:%02x:%02x.0' % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:           # noqa: BLE001 — older torch: ask the HIP runtime
        pass
    try:
        import ctypes
        hip = ctypes.CDLL('libamdhip64.so')
        buf = ctypes.create_string_buffer(32)
        if hip.hipDeviceGetPCIBusId(buf, 32, int(device_index)) == 0:
            return buf.value.decode().lower()
    except Exception:           # noqa: BLE001
        pass
    return None


def hwmon_of(pci):
    """hwmon directory of the card at that PCI address, or None."""
    if not pci:
        return None
    for card in glob.glob('/sys/class/drm/card*/device'):
        if os.path.basename(os.path.realpath(card)).lower() == pci.lower():
            hw = sorted(glob.glob(os.path.join(card, 'hwmon', 'hwmon*')))
            return hw[0] if hw else None
    return None


class BoardPower(threading.Thread):
    """with BoardPower(device_index) as bp: ... ; bp.summary() -> {'mean_W', 'max_W', 'cap_W', 'sclk_mean_MHz', 'sclk_min_MHz',
    'share_at_cap', 'samples', 'source'} or None when the box does not expose the files."""

    def __init__(self, device_index=0, period=0.05):
        super().__init__(daemon=True)
        self.period = period
        self.pci = pci_address(device_index)
        self.hw = hwmon_of(self.pci)
        self.samples = []
        self._stop_flag = False
        self._power = self._clk = None
        if self.hw:
            for name in ('power1_average', 'power1_input'):
                if _read(os.path.join(self.hw, name)).isdigit():
                    self._power = os.path.join(self.hw, name)
                    break
            if _read(os.path.join(self.hw, 'freq1_input')).isdigit():
                self._clk = os.path.join(self.hw, 'freq1_input')

    @property
    def available(self):
        return self._power is not None

    def one(self):
        pw = _read(self._power) if self._power else ''
        ck = _read(self._clk) if self._clk else ''
        return (int(pw) / 1e6 if pw.isdigit() else None, int(ck) / 1e6 if ck.isdigit() else None)

    def run(self):
        while not self._stop_flag:
            self.samples.append((time.time(),) + self.one())
            time.sleep(self.period)

    def __enter__(self):
        if self.available:
            self.start()
        return self

    def __exit__(self, *exc):
        self._stop_flag = True
        if self.is_alive():
            self.join()
        return False

    def cap_W(self):
        v = _read(os.path.join(self.hw, 'power1_cap')) if self.hw else ''
        return int(v) / 1e6 if v.isdigit() else None

    def summary(self, skip_frac=0.0):
        if not self.available or not self.samples:
            return None
        s = self.samples[int(len(self.samples) * skip_frac):]
        pw = [x[1] for x in s if x[1] is not None]
        ck = [x[2] for x in s if x[2] is not None]
        if not pw:
            return None
        out = {'mean_W': round(sum(pw) / len(pw), 1), 'max_W': round(max(pw), 1), 'cap_W': self.cap_W(), 'samples': len(pw),
               'source': f'sysfs hwmon of {self.pci} (average socket power, ~{int(1 / self.period)} Hz host thread)'}
        cap = out['cap_W']
        if cap:     # share of the samples within 3 % of the cap: how much of the region the power manager was the bound
            out['share_at_cap'] = round(sum(1 for v in pw if v >= 0.97 * cap) / len(pw), 3)
        if ck:
            out['sclk_mean_MHz'] = round(sum(ck) / len(ck), 0)
            out['sclk_min_MHz'] = round(min(ck), 0)
        return out
