"""``DataGenerator`` for the audio network (reference audio_network/data_generator.py:19-283): MFCC frames in,
WORD-level label rows out (each gesture class expands to its Italian words, 44 word classes incl. blank 43)."""
import os

import numpy as np

from ..datagen import BaseDataGenerator, CsvStore, SyntheticStore

# gesture class -> word ids (reference audio_network/data_generator.py:138-141)
class_2_words = {0: [0], 1: [1], 2: [2, 3], 3: [4], 4: [5, 6, 7], 5: [8, 9, 10], 6: [8, 11], 7: [12, 13], 8: [14, 15],
                 9: [16, 17], 10: [18, 19, 20, 21, 22], 11: [23], 12: [24, 25, 26], 13: [27], 14: [28, 11, 29],
                 15: [18, 30, 31, 32], 16: [33, 34], 17: [35, 36, 37], 18: [38], 19: [39, 40, 41, 13], 20: [40, 42],
                 21: [43]}


class DataGenerator(BaseDataGenerator):
    streams = (('the_input', 'audio', 'numfeats'),)
    model_json_name = "sp_ctc_lstm_model.json"
    model_weights_name = "sp_ctc_lstm_weights.h5"

    def __init__(self, minibatch_size, numfeats, maxlen, nb_classes, dataset, val_split=0.2,
                 absolute_max_sequence_len=150, data_root='../data', synthetic_files=None, seed=20131901,
                 word_level=True, rank=0, world=1):
        self.numfeats = numfeats
        self.word_level = word_level
        names = {'train': ('train_audio', 'training_oov.csv'), 'val': ('val_audio', 'validation.csv'),
                 'final': ('final_audio', 'validation.csv')}[dataset]
        self.in_audio_dir = os.path.join(data_root, names[0])
        if synthetic_files is None and os.path.isdir(self.in_audio_dir):
            store = CsvStore(self.in_audio_dir, None, os.path.join(data_root, names[1]))
        else:
            n = synthetic_files if synthetic_files is not None else 470
            # label rows are gesture classes 0..20; the word expansion below maps them into the 44-word space
            store = SyntheticStore(n, {'audio': (numfeats, 3.0)}, maxlen, 22 if word_level else nb_classes, seed=seed,
                                   lmax=20)
        self._setup(minibatch_size, maxlen, nb_classes, dataset, val_split, absolute_max_sequence_len, store, rank=rank, world=world)

    def sent_2_words(self, lab_seq):
        """Gesture-class label sequence -> word-level label sequence."""
        out = []
        for lab in lab_seq:
            out.extend(class_2_words[int(lab)])
        return np.asarray(out, dtype=np.float32)

    def expand_labels(self, lab_seq):
        if not self.word_level or lab_seq.shape[0] == 0:
            return lab_seq
        return self.sent_2_words(lab_seq)
