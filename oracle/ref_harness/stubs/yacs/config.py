"""Stand-in for yacs.config.CfgNode (absent here): attribute-style dict, used by src/config/EPN_options.py."""


class CfgNode(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v
