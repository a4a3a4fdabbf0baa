"""``DataGenerator`` for the fusion network (reference multimodal_fusion/data_generator.py:20-324)."""
import os

from ..datagen import BaseDataGenerator, CsvStore, SyntheticStore


class DataGenerator(BaseDataGenerator):
    streams = (('the_input_audio', 'audio', 'numfeats_speech'), ('the_input_skeletal', 'skeletal', 'numfeats_skeletal'))
    model_json_name = "multimodal_ctc_blstm_model.json"
    model_weights_name = "multimodal_ctc_blstm_weights.h5"

    def __init__(self, minibatch_size, numfeats_skeletal, numfeats_speech, maxlen, nb_classes, dataset, val_split=0.2,
                 absolute_max_sequence_len=35, data_root='../data', synthetic_files=None, seed=20131900, rank=0, world=1):
        self.numfeats_speech = numfeats_speech
        self.numfeats_skeletal = numfeats_skeletal
        names = {'train': ('train_audio', 'Training_set_skeletal.csv', 'training_oov.csv'),
                 'val': ('val_audio', 'Validation_set_skeletal.csv', 'validation.csv'),
                 'final': ('final_audio', 'final_set_skeletal.csv', 'validation.csv')}[dataset]
        self.in_audio_dir = os.path.join(data_root, names[0])
        self.in_file_skeletal = os.path.join(data_root, names[1])
        label_csv = os.path.join(data_root, names[2])
        if synthetic_files is None and os.path.isdir(self.in_audio_dir) and os.path.isfile(self.in_file_skeletal):
            store = CsvStore(self.in_audio_dir, self.in_file_skeletal, label_csv)
        else:
            n = synthetic_files if synthetic_files is not None else 470
            store = SyntheticStore(n, {'audio': (numfeats_speech, 3.0), 'skeletal': (numfeats_skeletal, 1.0)}, maxlen,
                                   nb_classes, seed=seed, lmax=min(20, absolute_max_sequence_len))
        self._setup(minibatch_size, maxlen, nb_classes, dataset, val_split, absolute_max_sequence_len, store, rank=rank, world=world)
