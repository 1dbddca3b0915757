"""Default EPN configuration (/root/reference/src/config/EPN_options.py:4-45) without the yacs dependency."""


class CN(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def get_default_cfg():
    cfg = CN()
    cfg.model = CN(model="enc_so3net", input_num=1024, output_num=32, search_radius=0.4, normalize_input=False, dropout_rate=0.0,
                   init_method="xavier", kpconv=False, kanchor=60, normals=False, flag="rotation", representation="quat",
                   pooling_method="max")
    cfg.MODEL = CN(num_in_channels=1, num_mid_channels=1, num_channels_R=1)
    cfg.train_lr = CN(init_lr=1e-3, lr_type="exp_decay", decay_rate=0.5, decay_step=10000)
    cfg.train_loss = CN(loss_type="soft", attention_loss_type="no_reg", margin=1.0, temperature=3, attention_margin=1.0,
                        attention_pretrain_step=3000, equi_alpha=0.0, eval=False)
    cfg.t_method_type = 2
    return cfg
