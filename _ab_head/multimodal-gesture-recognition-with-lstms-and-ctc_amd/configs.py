"""Named network configurations: the reference's true sizes and the BASELINE.json variants."""
from .spec import NetworkSpec

FUSION_OPT = {"lr": 1e-4, "decay": 1e-5, "clipvalue": 0.5, "maxnorm": 3.0}


def fusion_spec(numfeats_speech=39, numfeats_skeletal=20, nb_classes=22, h_audio=500, h_skeletal=300, h_fusion=100):
    """multimodal_fusion/multimodal.py:88-213 (frozen 2x BiLSTM encoders + residual, concat, BiLSTM(100), Dense)."""
    return NetworkSpec(
        streams=[
            {"name": "the_input_audio", "F": numfeats_speech, "noise": 0.5, "residual": True, "trainable": False,
             "layers": [{"H": h_audio, "dropout": 0.4, "name": "speech_blstm_1"},
                        {"H": h_audio, "dropout": 0.5, "name": "speech_blstm_2"}]},
            {"name": "the_input_skeletal", "F": numfeats_skeletal, "noise": 0.0, "residual": True, "trainable": False,
             "layers": [{"H": h_skeletal, "dropout": 0.6, "name": "skeletal_blstm_1"},
                        {"H": h_skeletal, "dropout": 0.6, "name": "skeletal_blstm_2"}]},
        ],
        fusion={"H": h_fusion, "dropout": 0.5, "name": "blstm_2"},
        head={"dropout": 0.5, "C": nb_classes, "dropout_name": "dropout_layer_3"},
        optimizer=dict(FUSION_OPT), name="multimodal_ctc_blstm")


def audio_spec(numfeats=39, nb_classes=44, h=500, layers=2):
    """audio_network/speech_lstm_ctc_words.py:42-132."""
    drops = [0.4, 0.5][:layers]
    return NetworkSpec(
        streams=[{"name": "the_input", "F": numfeats, "noise": 0.5, "residual": layers == 2, "trainable": True,
                  "layers": [{"H": h, "dropout": d, "name": "blstm_%d" % (i + 1)} for i, d in enumerate(drops)]}],
        fusion=None, head={"dropout": 0.5, "C": nb_classes, "dropout_name": "dropout_layer_1"},
        optimizer={"lr": 1e-4, "decay": 0.0, "clipvalue": 0.5, "maxnorm": 3.0}, name="sp_ctc_lstm")


def skeletal_spec(numfeats=20, nb_classes=22, h=300, layers=2):
    """skeletal_network/skeletal_lstm_ctc.py:298-394."""
    drops = [0.6, 0.6][:layers]
    return NetworkSpec(
        streams=[{"name": "the_input", "F": numfeats, "noise": 0.5, "residual": layers == 2, "trainable": True,
                  "layers": [{"H": h, "dropout": d, "name": "blstm_%d" % (i + 1)} for i, d in enumerate(drops)]}],
        fusion=None, head={"dropout": 0.6, "C": nb_classes, "dropout_name": "dropout_layer_1"},
        optimizer={"lr": 1e-4, "decay": 1e-5, "clipvalue": 0.5, "maxnorm": 3.0}, name="sk_ctc_lstm")


def early_fusion_spec(numfeats_speech=39, numfeats_skeletal=20, nb_classes=22, h=500):
    """early_fusion/early_multimodal.py:321-418: concat(noisy audio, noisy skeletal) -> 2x BiLSTM(500, .4) + add -> Dropout(.4)."""
    return NetworkSpec(
        streams=[{"name": "early_concat", "inputs": ["the_input_audio", "the_input_skeletal"],
                  "F": numfeats_speech + numfeats_skeletal, "noise": 0.5, "residual": True, "trainable": True,
                  "layers": [{"H": h, "dropout": 0.4, "name": "blstm_1"}, {"H": h, "dropout": 0.4, "name": "blstm_2"}]}],
        fusion=None, head={"dropout": 0.4, "C": nb_classes, "dropout_name": "dropout_layer_1"},
        optimizer={"lr": 1e-4, "decay": 1e-5, "clipvalue": 0.5, "maxnorm": 3.0}, name="early_multimodal")


def baseline_config(key):
    """BASELINE.json configs[] as (spec, B, T, Lmax)."""
    if key == "A":   # audio plumbing: 2-layer BiLSTM(128), 21 labels + blank
        return audio_spec(39, 22, 128, 2), 8, 200, 35
    if key == "A_ref":
        return audio_spec(39, 44, 500, 2), 8, 200, 150
    if key == "S":   # skeletal: BiLSTM(128)+CTC on 22-d feats
        return skeletal_spec(22, 22, 128, 1), 32, 1000, 28
    if key == "S_ref":
        return skeletal_spec(20, 22, 300, 2), 32, 1000, 28
    if key == "F":   # fusion at the reference's sizes (the metric's config)
        return fusion_spec(), 64, 1900, 35
    if key == "E":   # early fusion: same kernels with F = 59, H = 500
        return early_fusion_spec(), 16, 1900, 28
    if key == "F128":
        return fusion_spec(h_audio=128, h_skeletal=128, h_fusion=128), 64, 1900, 35
    raise KeyError(key)
