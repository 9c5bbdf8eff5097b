"""Re-serialise the reference's option file for BASELINE.json configs[0] (build container only):

    python tests/golden/make_golden_config.py

Reads /root/reference/options/test_videoswap/animal/2001_catheadturn_T05_Iter100/2001_catheadturn_T05_Iter100.yml with
this package's loader and writes the parsed dict as tests/golden/config1_options.json (the GPU box has no
/root/reference; tests/test_config.py checks the JSON against the YAML whenever the reference is present)."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from videoswap_amd.config import load_options  # noqa: E402

YML = '/root/reference/options/test_videoswap/animal/2001_catheadturn_T05_Iter100/2001_catheadturn_T05_Iter100.yml'

if __name__ == '__main__':
    with open(os.path.join(HERE, 'config1_options.json'), 'w') as f:
        json.dump({'source': YML, 'options': load_options(YML)}, f, indent=1, sort_keys=True)
    print('wrote config1_options.json')
