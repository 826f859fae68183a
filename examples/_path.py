"""Put the drop-in packages (models/, utils/, nsgp/) of this repo on sys.path, like
`export PYTHONPATH=<repo>/nonstationary-precip_amd` in INTEGRATION.md."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'nonstationary-precip_amd')
if PKG not in sys.path:
    sys.path.insert(0, PKG)
