"""Configuration knobs of the hot path - same names and default values as the reference's
``mycode/config.py`` (``cfg`` there is an ``EasyDict``; here a plain attribute dict, no
third-party dependency).  Only knobs the seq2seq path reads are kept; citations are
config.py line numbers in /root/reference/mycode/.
"""


class AttrDict(dict):
    """dict with attribute access (read, write, delete)."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        try:
            del self[key]
        except KeyError:
            raise AttributeError(key)


def default_config():
    c = AttrDict()
    c.training_epochs = 30            # :6
    c.use_xyz = True                  # :8
    c.process_in_seconds = True       # :12
    c.batch_size = 32                 # :13
    c.fps = 30                        # :14
    c.predict_len = 10                # :21
    c.running_length = 10             # :22  encoder length T_in (seconds)
    c.predict_step = 10               # :23  decoder length T_out (seconds)
    c.add_residual_link = False       # :38  (FoV_seq2seq_no_teac_forc.py)
    c.has_reconstruct_loss = False    # :40  (not built)
    c.LEARNING_RATE = 1e-5            # :50  (raw-TF path)
    c.lr_epoch_step = 10              # :51
    c.clip_gradient = True            # :52
    c.add_xyz_sum1 = False            # :54
    c.shuffle_data = False            # :61
    c.stateful_across_batch = False   # :62
    c.dropout_rate = 0.3              # :64
    c.conv_kernel_size = 5            # :65
    c.predict_mean_var = False        # :69  (the others-mixing model needs True)
    c.sample_and_refeed = True        # :70
    c.use_GMM = True                  # :71  mixture-density head of lstm.py (training.TFLSTMTrainer.head_kind_of)
    c.input_mean_var = False          # :74
    c.teacher_forcing = False         # :75
    c.use_one_hot = False             # :76
    c.use_overlapping_chunks = True   # :86
    c.data_chunk_stride = 10          # :89
    c.dilation_rate = 1               # :105
    c.use_saliency = False            # :107
    c.cut_data_head = False           # :112
    c.purelly_testing = False         # :113
    c.time_shift = False              # :120
    c.enc_last_out_as_dec_in = False  # :121
    c.embed_frame_state_enc2dec = False   # :127 (not built)
    c.rescale_input = False           # :130
    c.include_time_ind = False        # :131
    # not in the reference: gate activation of the LSTM cell.  Keras < 2.3 defaults to
    # 'hard_sigmoid'; BASELINE.json's north_star names 'sigmoid'.
    c.recurrent_activation = "sigmoid"
    return c


cfg = default_config()
