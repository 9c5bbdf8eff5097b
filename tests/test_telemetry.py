"""videoswap_amd/telemetry.py: the board-power sampler of bench.py (`readings.board_power`) on a fake sysfs tree — device matched by
PCI address, average / cap / share of samples at the cap, and `None` (never an exception, never another tenant's card) when the box
exposes nothing for this device."""
import os
import time

from videoswap_amd import telemetry


def _fake_tree(tmp_path, cards):
    """cards: {pci: (power_uW, freq_Hz, cap_uW)} -> glob pattern for /sys/class/drm/card*/device"""
    for i, (pci, (pw, fq, cap)) in enumerate(cards.items()):
        real = tmp_path / 'devices' / pci
        hw = real / 'hwmon' / f'hwmon{i + 3}'
        hw.mkdir(parents=True)
        (hw / 'power1_average').write_text(f'{pw}\n')
        (hw / 'freq1_input').write_text(f'{fq}\n')
        (hw / 'power1_cap').write_text(f'{cap}\n')
        card = tmp_path / 'drm' / f'card{i}'
        card.mkdir(parents=True)
        os.symlink(real, card / 'device')
    return str(tmp_path / 'drm' / 'card*' / 'device')


def test_board_power_reads_the_matching_card_only(tmp_path, monkeypatch):
    pattern = _fake_tree(tmp_path, {'0000:72:00.0': (1_390_000_000, 1_640_000_000, 1_400_000_000),
                                    '0000:05:00.0': (300_000_000, 2_400_000_000, 1_400_000_000)})
    real_glob = telemetry.glob.glob
    monkeypatch.setattr(telemetry.glob, 'glob', lambda p: real_glob(pattern if p == '/sys/class/drm/card*/device' else p))
    monkeypatch.setattr(telemetry, 'pci_address', lambda i=0: '0000:72:00.0')
    with telemetry.BoardPower(0, period=0.005) as bp:
        time.sleep(0.06)
    s = bp.summary()
    assert bp.available and s is not None
    assert s['mean_W'] == 1390.0 and s['max_W'] == 1390.0 and s['cap_W'] == 1400.0
    assert s['share_at_cap'] == 1.0                 # 1 390 W is within 3 % of the cap
    assert s['sclk_mean_MHz'] == 1640.0 and s['samples'] >= 3
    assert '0000:72:00.0' in s['source']


def test_board_power_is_none_without_a_matching_card(tmp_path, monkeypatch):
    pattern = _fake_tree(tmp_path, {'0000:05:00.0': (300_000_000, 2_400_000_000, 1_400_000_000)})
    real_glob = telemetry.glob.glob
    monkeypatch.setattr(telemetry.glob, 'glob', lambda p: real_glob(pattern if p == '/sys/class/drm/card*/device' else p))
    monkeypatch.setattr(telemetry, 'pci_address', lambda i=0: '0000:72:00.0')      # another tenant's card is the only one visible
    with telemetry.BoardPower(0) as bp:
        pass
    assert not bp.available and bp.summary() is None


def test_board_power_without_a_device_never_raises():
    with telemetry.BoardPower(-1) as bp:            # CPU plumbing runs of bench.py
        pass
    assert bp.summary() is None
