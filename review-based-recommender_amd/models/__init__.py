"""nn.Module mirrors of the reference's models/{deepconn,narre,dual_att} packages."""
