"""`promonet_amd.patch(promonet)`: swap the HIP engine into an installed
reference package, the literal drop-in.

After patching, `promonet.synthesize.from_features(..., gpu=N)`,
`promonet.model.Generator()` and `promonet.preprocess.spectrogram.from_audio`
/ `promonet.preprocess.loudness.from_audio` run on libpromonet_hip.so.
"""
import promonet_amd


def patch(promonet):
    """Monkey-patch the reference package object in place and return it."""
    for name in dir(promonet_amd.config):
        if name.isupper() and hasattr(promonet, name) and name not in (
                'ASSETS_DIR', 'COMPUTE_DTYPE'):
            value = getattr(promonet, name)
            if getattr(promonet_amd, name) != value:
                promonet_amd.configure(**{name: value})

    promonet.model.HiFiGAN = promonet_amd.model.HiFiGAN
    promonet.model.FARGAN = promonet_amd.model.FARGAN
    promonet.model.Generator = promonet_amd.model.Generator
    for name in (
        'from_features', 'from_file', 'from_file_to_file',
        'from_files_to_files', 'generate'
    ):
        setattr(promonet.synthesize, name,
                getattr(promonet_amd.synthesize, name))
        if hasattr(promonet.synthesize, 'core'):
            setattr(promonet.synthesize.core, name,
                    getattr(promonet_amd.synthesize, name))
    promonet.preprocess.spectrogram.from_audio = \
        promonet_amd.preprocess.spectrogram.from_audio
    promonet.preprocess.spectrogram.linear_to_mel = \
        promonet_amd.preprocess.spectrogram.linear_to_mel
    promonet.preprocess.loudness.from_audio = \
        promonet_amd.preprocess.loudness.from_audio
    return promonet
